#!/bin/bash
# round 3: the 8 x 1080p -> 4K tiled scene (tests/scenes.py cfg2_scene: Tiles, rescale + blend) per INPUT format and per OUTPUT format,
# one context, one frame in flight, smr_render_layouts in a loop (HIP events over 100 frames)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY'
import numpy as np
from smelter_amd import hip
from tests import scenes
ctx = hip.Context(0)
w, h, W, H, n = 1920, 1080, 3840, 2160, 8
layouts, res = scenes.cfg2_scene(w, h, W, H, n)
rng = np.random.default_rng(3)
def frames(fmt):
    out = []
    for i in range(n):
        y = rng.integers(16, 236, (h, w), dtype=np.uint8)
        if fmt == "yuv420": out.append(ctx.frame(hip.FRAME_PLANAR_YUV420, w, h, [y, rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8), rng.integers(16, 241, (h // 2, w // 2), dtype=np.uint8)]))
        elif fmt == "nv12": out.append(ctx.frame(hip.FRAME_NV12, w, h, [y, rng.integers(16, 241, (h // 2, w // 2, 2), dtype=np.uint8)]))
        elif fmt == "yuv422": out.append(ctx.frame(hip.FRAME_PLANAR_YUV422, w, h, [y, rng.integers(16, 241, (h, w // 2), dtype=np.uint8), rng.integers(16, 241, (h, w // 2), dtype=np.uint8)]))
        elif fmt == "yuv444": out.append(ctx.frame(hip.FRAME_PLANAR_YUV444, w, h, [y, rng.integers(16, 241, (h, w), dtype=np.uint8), rng.integers(16, 241, (h, w), dtype=np.uint8)]))
        elif fmt == "uyvy": out.append(ctx.frame(hip.FRAME_UYVY422, w, h, [rng.integers(16, 236, (h, w // 2, 4), dtype=np.uint8)]))
        elif fmt == "bgra": out.append(ctx.frame(hip.FRAME_BGRA, w, h, [rng.integers(0, 256, (h, w, 4), dtype=np.uint8)]))
    return out
def run(src, out):
    for _ in range(10): ctx.render_layouts(layouts, src, W, H, out=out)
    ctx.sync(); ctx.timer_start()
    for _ in range(100): ctx.render_layouts(layouts, src, W, H, out=out)
    return ctx.timer_stop() * 1000 / 100
o420 = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
for fmt in ("yuv420", "nv12", "yuv422", "yuv444", "uyvy", "bgra"):
    us = run(frames(fmt), o420)
    print(f"inputs {fmt:7s} -> yuv420 out: {us:7.1f} us per frame ({1e6 / us:7.0f} frames/s)")
src = frames("yuv420")
for name, f in (("nv12", hip.FRAME_NV12), ("yuv422", hip.FRAME_PLANAR_YUV422), ("yuv444", hip.FRAME_PLANAR_YUV444)):
    us = run(src, ctx.frame(f, W, H))
    print(f"inputs yuv420  -> {name:6s} out: {us:7.1f} us per frame ({1e6 / us:7.0f} frames/s)")
PY
