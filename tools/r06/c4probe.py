#!/usr/bin/env python3
"""configs[4] frozen in mid-transition: one lane renders the SAME presentation time (250 ms into a 500 ms transition) N times, so that the
kernels of a frame in motion can be read alone (stage timers, or rocprofv3 --kernel-trace around this script).  Laboratory builds take
SMR_ABLATE: bits 8.. of it reach the compositor (32 << 8 no sampled tiles, 16 << 8 no composited tiles, 64 << 8 no copy tiles).
usage: python tools/r06/c4probe.py [frames] [fraction of the transition, default 0.5]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from smelter_amd import hip, synth
from smelter_amd.renderer import Renderer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
FRAC = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ctx = hip.Context(0)
r = Renderer(ctx, stream_fallback_timeout_s=3600.0)
frames = {}
for i in range(16):
    r.register_input(f"input_{i}")
    y, u, v = synth.test_input(i, 1920, 1080, noise_seed=1234 + i, shift=0)
    frames[f"input_{i}"] = ctx.frame(hip.FRAME_PLANAR_YUV420, 1920, 1080, [y, u, v])
r.register_shader("soften")
r.update_scene("out", 3840, 2160, synth.animated_grid_scene(16, 0, 960, 540, 500, 3.0))
fs = r.make_frame_set(frames)
NS = 1_000_000_000
r.render_packed(0, fs)
r.render_packed(NS // 60, fs)
r.update_scene("out", 3840, 2160, synth.animated_grid_scene(16, 1, 960, 540, 500, 3.0))
t0 = 2 * NS // 60
r.render_packed(t0, fs)                       # the transition starts at the first frame after the update
pts = t0 + int(FRAC * 0.5 * NS)
for _ in range(8):
    r.render_packed(pts, fs)
ctx.sync()
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(N):
    r.render_packed(pts, fs)
ctx.sync()
prof = ctx.profile_read(); ctx.profile_enable(False)
print(json.dumps({"ablate": os.environ.get("SMR_ABLATE"), "frac": FRAC,
                  "stages_us": {k: [round(1000.0 * ms / n, 2), n] for k, (ms, n) in prof.items() if n}}))
