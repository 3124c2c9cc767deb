"""How long the host spends enqueuing one frame (smr_renderer_render on the benchmark scene, no synchronisation inside the loop)
against the GPU's time per frame: tells whether the pipelined rate is bound by the host thread or by the device.
python tools/host_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from smelter_amd import _ffi, hip, synth  # noqa: E402
from smelter_amd.renderer import Renderer  # noqa: E402


def main():
    ctx = hip.Context(0)
    lanes = [hip.Context(0) for _ in range(2)]
    r = Renderer(ctx, stream_fallback_timeout_s=3600.0, lanes=lanes)
    for i in range(bench.N_IN):
        r.register_input(f"input_{i}")
    atlas, glyphs = synth.label_glyphs("CAM 3 LIVE", 3)
    for node in r.update_scene("out", bench.OUT_W, bench.OUT_H, bench.scene_json()):
        if node.kind == _ffi.NODE_TEXT:
            r.set_text("out", node.index, glyphs, atlas)
    ring = bench.make_inputs(ctx, hip, 4, list(range(bench.N_IN)))
    sets = [r.make_frame_set({f"input_{i}": row[i] for i in range(bench.N_IN)}) for row in ring]
    ns = 1_000_000_000 // 60
    for s in range(50):
        r.render_packed(s * ns, sets[s % 4])
    r.sync()
    n = 3000
    t0 = time.perf_counter()
    for s in range(n):
        r.render_packed((50 + s) * ns, sets[s % 4])
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"host enqueue {1e6 * (t1 - t0) / n:.1f} us per frame; with the device drained {1e6 * (t2 - t0) / n:.1f} us per frame "
          f"({n / (t2 - t0):.0f} frames/s)")


if __name__ == "__main__":
    main()
