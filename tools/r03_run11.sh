#!/bin/bash
# round 3: smr_rgba_to_frame at 3840x2160 per output format, one-launch kernel against the three-pass kernels (SMR_CONVERT_GENERAL=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY'
import os, numpy as np
from smelter_amd import hip
ctx = hip.Context(0)
w, h = 3840, 2160
node = ctx.surface_from(np.random.default_rng(1).integers(0, 256, (h, w, 4), dtype=np.uint8))
for name, fmt in (("yuv420", hip.FRAME_PLANAR_YUV420), ("yuv422", hip.FRAME_PLANAR_YUV422), ("yuv444", hip.FRAME_PLANAR_YUV444), ("nv12", hip.FRAME_NV12)):
    out = ctx.frame(fmt, w, h)
    res = []
    for general in (False, True):
        if general: os.environ["SMR_CONVERT_GENERAL"] = "1"
        for _ in range(5): ctx.rgba_to_frame(node, fmt, out)
        ctx.sync(); ctx.timer_start()
        for _ in range(50): ctx.rgba_to_frame(node, fmt, out)
        res.append(ctx.timer_stop() * 1000 / 50)
        os.environ.pop("SMR_CONVERT_GENERAL", None)
    print(f"{name:8s} one launch {res[0]:6.1f} us   three passes {res[1]:6.1f} us per 3840x2160 frame")
PY
