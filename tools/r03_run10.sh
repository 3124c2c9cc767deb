#!/bin/bash
# round 3: smr_frame_to_rgba per format (1920x1080), batched structure-aware kernel against the general kernels (SMR_CONVERT_GENERAL=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY'
import os, numpy as np, time
from smelter_amd import hip
ctx = hip.Context(0)
w, h = 1920, 1080
rng = np.random.default_rng(1)
def frames():
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    out = {}
    for name, fmt, cw, ch in (("yuv420", hip.FRAME_PLANAR_YUV420, w // 2, h // 2), ("yuv422", hip.FRAME_PLANAR_YUV422, w // 2, h), ("yuv444", hip.FRAME_PLANAR_YUV444, w, h)):
        out[name] = ctx.frame(fmt, w, h, [y, rng.integers(0, 256, (ch, cw), dtype=np.uint8), rng.integers(0, 256, (ch, cw), dtype=np.uint8)])
    out["nv12"] = ctx.frame(hip.FRAME_NV12, w, h, [y, rng.integers(0, 256, (h // 2, w // 2, 2), dtype=np.uint8)])
    out["uyvy"] = ctx.frame(hip.FRAME_UYVY422, w, h, [rng.integers(0, 256, (h, w // 2, 4), dtype=np.uint8)])
    return out
F = frames()
node = ctx.surface(w, h)
for name, f in F.items():
    res = []
    for general in (False, True):
        if general: os.environ["SMR_CONVERT_GENERAL"] = "1"
        for _ in range(5): ctx.frame_to_rgba(f, node)
        ctx.sync(); ctx.timer_start()
        for _ in range(50): ctx.frame_to_rgba(f, node)
        res.append(ctx.timer_stop() * 1000 / 50)
        os.environ.pop("SMR_CONVERT_GENERAL", None)
    print(f"{name:8s} batched kernel {res[0]:6.1f} us   general kernel {res[1]:6.1f} us per 1920x1080 frame")
PY
