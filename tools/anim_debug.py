"""Debug driver for the animated grid scene (bench.py --config 4): one renderer, scene updates every 45 frames, a watchdog thread
that reports the step the main thread is in and the process CPU time (host spin vs waiting on the GPU)."""
import faulthandler
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("DEBUG_TIMEOUT", "70")), exit=True)

from smelter_amd import hip, synth  # noqa: E402
from smelter_amd.renderer import Renderer  # noqa: E402

N, W, H = 16, 3840, 2160
state = {"step": -1, "phase": -1, "what": ""}


def watchdog():
    while True:
        time.sleep(3)
        t = os.times()
        print(f"[wd] phase {state['phase']} step {state['step']} {state['what']} cpu user {t.user:.1f} sys {t.system:.1f}", flush=True)


threading.Thread(target=watchdog, daemon=True).start()
ctx = hip.Context(0)
r = Renderer(ctx, stream_fallback_timeout_s=3600.0)
frames = {}
for i in range(N):
    y, u, v = synth.test_input(i, 1920, 1080, noise_seed=i)
    r.register_input(f"input_{i}")
    frames[f"input_{i}"] = ctx.frame(hip.FRAME_PLANAR_YUV420, 1920, 1080, [y, u, v])
r.register_shader("soften")
fs = r.make_frame_set(frames)
FRAME_NS = 1_000_000_000 // 60
profile = os.environ.get("DEBUG_PROFILE") == "1"
for phase in range(int(os.environ.get("DEBUG_PHASES", "4"))):
    state["phase"] = phase
    if profile and phase == 2:
        ctx.profile_reset()
        ctx.profile_enable(True)
    t0 = time.perf_counter()
    worst = 0.0
    for s in range(int(os.environ.get("DEBUG_STEPS", "200"))):
        state["step"] = s
        t1 = time.perf_counter()
        if s % 45 == 0:
            state["what"] = "update"
            r.update_scene("out", W, H, synth.animated_grid_scene(N, s // 45))
        state["what"] = "render"
        # DEBUG_REWIND=1: pts restarts at 0 in every phase (outside the contract: transitions then extrapolate / go NaN, which
        # must stay cheap and must not hang anything)
        n_steps = int(os.environ.get("DEBUG_STEPS", "200"))
        r.render_packed((s if os.environ.get("DEBUG_REWIND") == "1" else phase * n_steps + s) * FRAME_NS, fs)
        if os.environ.get("DEBUG_SYNC") == "1":
            state["what"] = "sync"
            ctx.sync()
        dt = time.perf_counter() - t1
        if dt > worst:
            worst = dt
            print(f"  phase {phase} step {s}: {dt * 1e3:.2f} ms", flush=True)
    state["what"] = "phase sync"
    ctx.sync()
    print(f"phase {phase}: {time.perf_counter() - t0:.3f} s", flush=True)
print("done", flush=True)
