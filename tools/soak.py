import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from smelter_amd import hip, synth
from smelter_amd.renderer import Renderer
ctx = hip.Context(0)
r = Renderer(ctx)
n = 6
frames = {}
for i in range(n):
    y,u,v = synth.test_input(i, 1280, 720, noise_seed=i)
    r.register_input(f"in{i}")
    frames[f"in{i}"] = ctx.frame(hip.FRAME_PLANAR_YUV420, 1280, 720, [y,u,v])
def scene(k):
    order = list(range(n)); rot = k % n; order = order[rot:] + order[:rot]
    kids = [{"type": "input_stream", "input_id": f"in{i}", "id": f"s{i}"} for i in order[: 2 + k % (n-1)]]
    return {"type": "tiles", "id": "t", "margin": 4, "background_color": "#101010FF", "transition": {"duration_ms": 300, "easing_function": {"function_name": "bounce"}}, "children": kids}
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
pts = 0.0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    r.update_scene("out", 1920, 1080, scene(k))
    for j in range(5):
        pts += 1/60
        out = r.render(pts, frames, {f"in{i}": pts for i in range(n)})["out"]
ctx.sync()
dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
y = out.download()[0]
print("ok frames", 1500, "fps", round(1500/dt), "mem delta MB", (free0-free1)/1e6, "mean Y", float(y.mean()))
