"""Times one 1080p input laid out into very small rects (a tile growing from nothing in a transition): the whole
smr_render_layouts call per destination size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smelter_amd import hip, synth  # noqa: E402
from smelter_amd.scene import Layout  # noqa: E402

ctx = hip.Context(0)
y, u, v = synth.test_input(0, 1920, 1080, noise_seed=1)
frame = ctx.frame(hip.FRAME_PLANAR_YUV420, 1920, 1080, [y, u, v])
out = ctx.frame(hip.FRAME_PLANAR_YUV420, 1920, 1080)
for w, h in [(1, 1), (2, 1), (4, 2), (16, 9), (64, 36), (240, 135), (480, 270), (960, 540)]:
    L = [Layout(10.0, 10.0, float(w), float(h), 0.0, [0.0] * 4, 0, 0, [0.0] * 4, [0.0] * 4, 0.0, [0.0, 0.0, 1920.0, 1080.0], 0.0, [])]
    ctx.render_layouts(L, [frame], 1920, 1080, out=out)
    ctx.sync()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ctx.render_layouts(L, [frame], 1920, 1080, out=out)
    ctx.sync()
    print(f"{w}x{h}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame", flush=True)
