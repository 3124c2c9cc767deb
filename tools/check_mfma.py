"""Matrix-core resampler (k_ingest_mfma) against the f32 kernel and the CPU oracle on one input -> tile resample.

python tools/check_mfma.py [iw ih dw dh] ...   prints max |diff| and % identical bytes for MFMA vs oracle, VALU vs oracle,
and the mean launch time of both kernels (HIP events through smr_profile_*)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402  (checker only)
from smelter_amd import hip  # noqa: E402
from tests import scenes  # noqa: E402


def run(ctx, impl, frames, crops, dsts, reps):
    ctx.set_ingest_impl(impl)
    ctx.ingest_resample_batch(frames, crops, dsts)
    ctx.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.ingest_resample_batch(frames, crops, dsts)
    ctx.sync()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    ms, n = prof["fused_ingest_resample"]
    return [d.download() for d in dsts], (1000.0 * ms / max(n, 1)), n


def main():
    geoms = [(1920, 1080, 1280, 720, 8), (1920, 1080, 960, 540, 4), (640, 360, 426, 240, 2), (322, 182, 250, 141, 1), (3840, 2160, 1280, 720, 8)]
    if len(sys.argv) >= 5:
        geoms = [tuple(int(a) for a in sys.argv[1:5]) + (int(sys.argv[5]) if len(sys.argv) > 5 else 1,)]
    ctx = hip.Context(0)
    for iw, ih, dw, dh, n in geoms:
        frames, nodes = [], []
        for i in range(n):
            y, u, v = scenes.test_input(i, iw, ih, noise_seed=1234 + i)
            frames.append(ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, [y, u, v]))
            if i < 2:
                nodes.append(orc.planar_yuv_to_rgba(y, u, v, iw, ih))
        crops = [(0.0, 0.0, float(iw), float(ih))] * n
        dsts = [ctx.surface(dw, dh) for _ in range(n)]
        reps = 20
        got_m, us_m, n_m = run(ctx, hip.INGEST_MFMA_F16, frames, crops, dsts, reps)
        got_v, us_v, n_v = run(ctx, hip.INGEST_VALU_F32, frames, crops, dsts, reps)
        print(f"{iw}x{ih} -> {dw}x{dh} x{n}: mfma {us_m:.1f} us/launch ({n_m} launches), valu {us_v:.1f} us/launch ({n_v})")
        for i in range(min(n, 2)):
            t0 = time.time()
            _, want = orc.resample(nodes[i], crops[i], dw, dh, omp=True)
            for name, got in (("mfma", got_m[i]), ("valu", got_v[i])):
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                print(f"   input {i} {name}: max {d.max()}  identical {100.0 * (d == 0).mean():.3f}%  >1: {(d > 1).sum()}")
            dm = np.abs(got_m[i].astype(np.int32) - got_v[i].astype(np.int32))
            print(f"   input {i} mfma vs valu: max {dm.max()} identical {100.0 * (dm == 0).mean():.3f}%  (oracle {time.time() - t0:.1f}s)")
        for f in frames:
            f.destroy()
        for d in dsts:
            d.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
