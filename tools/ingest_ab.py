"""A/B of the three routes of wave A on BASELINE configs[2] (8 x 1920x1080 4:2:0 -> 8 x 1280x720 tiles) and on the north-star target
(8 x 3840x2160 -> 8 x 1280x720):

    f32_valu   k_ingest_resample                    every pass as the WGSL writes it (bit-identical to the pass-per-launch kernels)
    fused      k_ingest_wave on the planes          opt-in: conversion fused into the matrix-core resampler (within one code per stage)
    node       k_yuv420_to_rgba + k_ingest_wave     the default: exact converter into node textures, matrix-core resampler on them

per content class: mean time of wave A per call (HIP events around each launch, smr_profile_*: converter + resampler), max |difference| and
share of identical bytes against the CPU oracle on the first two inputs.  Writes gpurun_out/r04_ingest_ab.json.

python tools/ingest_ab.py [reps] [--4k] [--impls node,fused,f32_valu] [--contents bench,smooth,noise]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402  (checker only)
from smelter_amd import hip  # noqa: E402
from tests import scenes  # noqa: E402


def content(kind, i, w, h):
    if kind == "bench":
        return scenes.test_input(i, w, h, noise_seed=1234 + i)
    rng = np.random.default_rng(50 + i)
    if kind == "noise":
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
    prof = ctx.profile_read()
    ms, n = prof["fused_ingest_resample"]
    ms += prof["ingest"][0]  # (the node route's converter launch)
    ctx.profile_enable(False)
    return [d.download() for d in dsts], 1000.0 * ms / max(n, 1), n // max(reps, 1)


def stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"max_lsb": int(d.max()), "identical_pct": round(100.0 * float((d == 0).mean()), 4), "bytes_off_by_more_than_1": int((d > 1).sum())}


IMPLS = {"f32_valu": hip.INGEST_VALU_F32, "node": hip.INGEST_AUTO, "fused": hip.INGEST_MFMA_F16_FUSED}


def main():
    argv = sys.argv[1:]
    reps = int(argv[0]) if argv and argv[0].isdigit() else 50
    four_k = "--4k" in argv
    impls = argv[argv.index("--impls") + 1].split(",") if "--impls" in argv else ["f32_valu", "fused", "node"]
    kinds = argv[argv.index("--contents") + 1].split(",") if "--contents" in argv else ["bench", "smooth", "noise"]
    iw, ih = (3840, 2160) if four_k else (1920, 1080)
    dw, dh, n = 1280, 720, 8
    ctx = hip.Context(0)
    out = {"workload": f"{n} x {iw}x{ih} YUV420 -> {n} x {dw}x{dh} RGBA8 tiles", "reps": reps, "rows": []}
    for kind in kinds:
        planes = [content(kind, i, iw, ih) for i in range(n)]
        frames = [ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
        crops = [(0.0, 0.0, float(iw), float(ih))] * n
        dsts = [ctx.surface(dw, dh) for _ in range(n)]
        want = [orc.resample(orc.planar_yuv_to_rgba(*planes[i], iw, ih, omp=True), crops[i], dw, dh, omp=True)[1] for i in range(2)]
        row = {"content": kind}
        for name in impls:
            got, us, launches = run(ctx, IMPLS[name], frames, crops, dsts, reps)
            row[name] = {"us_per_launch": round(us, 2), "launches_per_call": launches, "vs_oracle": stats(np.stack(got[:2]), np.stack(want))}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
        for f in frames:
            f.destroy()
        for d in dsts:
            d.destroy()
    ctx.close()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r04_ingest_ab%s.json" % ("_4k" if four_k else ""), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
