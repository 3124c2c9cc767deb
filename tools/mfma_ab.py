"""The A/B the north star asks for, on BASELINE configs[2]'s wave A (8 x 1920x1080 4:2:0 -> 8 x 1280x720 linear-light Lanczos3 tiles):

    f32 VALU   k_ingest_resample  every pass as the WGSL writes it, per-operation f32 rounding (bit-identical to the oracle)
    MFMA f16   k_ingest_mfma      the two Lanczos passes as banded GEMMs on v_mfma_f32_16x16x32_f16 (f16 hi/lo pairs, f32 accumulate)

per content class: mean launch time over `reps` launches (HIP events around each launch, smr_profile_*), max |difference| and share of
identical bytes against the CPU oracle on the first two inputs, and MFMA against VALU on all eight.  Writes gpurun_out/r02_mfma_ab.json.

python tools/mfma_ab.py [reps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402  (checker only)
from smelter_amd import hip  # noqa: E402
from tests import scenes  # noqa: E402


def content(kind, i, w, h):
    if kind == "bench (synthetic camera + noise)":
        return scenes.test_input(i, w, h, noise_seed=1234 + i)
    rng = np.random.default_rng(50 + i)
    if kind == "white noise":
        return (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8),
                rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8))
    yy, xx = np.mgrid[0:h, 0:w]
    y = (16 + 219 * (0.5 + 0.5 * np.sin(xx / (37.0 + i) + yy / 53.0))).astype(np.uint8)  # smooth gradients
    u = (128 + 100 * np.sin(xx[::2, ::2] / 91.0)).astype(np.uint8)
    v = (128 + 100 * np.cos(yy[::2, ::2] / 67.0)).astype(np.uint8)
    return y, u, v


def run(ctx, impl, frames, crops, dsts, reps):
    ctx.set_ingest_impl(impl)
    ctx.ingest_resample_batch(frames, crops, dsts)
    ctx.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.ingest_resample_batch(frames, crops, dsts)
    ctx.sync()
    ms, n = ctx.profile_read()["fused_ingest_resample"]
    ctx.profile_enable(False)
    return [d.download() for d in dsts], 1000.0 * ms / max(n, 1)


def stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"max_lsb": int(d.max()), "identical_pct": round(100.0 * float((d == 0).mean()), 4), "bytes_off_by_more_than_1": int((d > 1).sum())}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    iw, ih, dw, dh, n = 1920, 1080, 1280, 720, 8
    ctx = hip.Context(0)
    out = {"workload": f"{n} x {iw}x{ih} YUV420 -> {n} x {dw}x{dh} RGBA8 tiles (configs[2] wave A)", "reps": reps, "rows": []}
    for kind in ("bench (synthetic camera + noise)", "smooth gradients", "white noise"):
        planes = [content(kind, i, iw, ih) for i in range(n)]
        frames = [ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
        crops = [(0.0, 0.0, float(iw), float(ih))] * n
        dsts = [ctx.surface(dw, dh) for _ in range(n)]
        got_v, us_v = run(ctx, hip.INGEST_VALU_F32, frames, crops, dsts, reps)
        got_m, us_m = run(ctx, hip.INGEST_MFMA_F16, frames, crops, dsts, reps)
        want = [orc.resample(orc.planar_yuv_to_rgba(*planes[i], iw, ih), crops[i], dw, dh, omp=True)[1] for i in range(2)]
        row = {"content": kind,
               "f32_valu": {"us_per_launch": round(us_v, 2), "vs_oracle": stats(np.stack(got_v[:2]), np.stack(want))},
               "mfma_f16": {"us_per_launch": round(us_m, 2), "vs_oracle": stats(np.stack(got_m[:2]), np.stack(want)),
                            "vs_f32_valu_all_inputs": stats(np.stack(got_m), np.stack(got_v))}}
        out["rows"].append(row)
        print(json.dumps(row))
        for f in frames:
            f.destroy()
        for d in dsts:
            d.destroy()
    ctx.close()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r02_mfma_ab.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
