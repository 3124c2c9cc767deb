"""A soak run of the renderer on the benchmark scene: minutes of frames on two lanes with everything a live host does in between —
scene updates that start transitions (tiles reordered by id, an input leaving and coming back), inputs unregistered and registered
again, text drawn by the library's font book at every update — while watching what must not move:
  * device memory (hipMemGetInfo through torch) over the whole run and the host's resident set over its last two thirds (the first read-backs
    grow the runtime's staging pool once: ~200 MB over the first few hundred downloads, nothing after — measured with frames only, updates only
    and downloads only): no growth,
  * the picture: the frame of a scene at rest, for the same input set, has the same checksum every time it comes round,
  * the error channel: no call fails.
python tools/soak.py [--seconds 150] [--variety]          prints one JSON line
--variety: three lanes instead of two, 720p NV12 frames on two of the inputs (the block converter's pitch-filling rows), and a second output —
1080p NV12 — whose scene (synth.animated_grid_scene: a grid in a cubic-bezier transition under a gaussian-blur shader layer, plus an image and
a text node) is replaced every time as well; both outputs' pictures at rest are compared."""
import argparse
import json
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from smelter_amd import hip  # noqa: E402
from smelter_amd.renderer import Renderer  # noqa: E402


def scenes():
    a = bench.scene_json()
    for i, kid in enumerate(a["children"]):
        kid["id"] = f"cell_{i}"
    a["id"] = "grid"
    a["transition"] = {"duration_ms": 300}
    b = json.loads(json.dumps(a))
    b["children"] = b["children"][3:] + b["children"][:2]  # reordered, one input gone: tiles move and resize by id
    return a, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--variety", action="store_true")
    args = ap.parse_args()
    import psutil
    import torch
    ctx = hip.Context(0)
    r = Renderer(ctx, stream_fallback_timeout_s=1e9, lanes=[hip.Context(0) for _ in range(2 if args.variety else 1)])  # (the ring's frames carry pts 0: never stale)
    book = bench.font_book()
    if book is not None:
        r.set_fontbook(book)
    for i in range(bench.N_IN):
        r.register_input(f"input_{i}")
    ring = bench.make_inputs(ctx, hip, 4, list(range(bench.N_IN)))
    second = None
    if args.variety:
        import numpy as np
        from smelter_amd import synth
        rng = np.random.default_rng(7)
        for s_, row in enumerate(ring):  # inputs 6 and 7 become 1280 x 720 NV12
            for i in (6, 7):
                y = rng.integers(16, 236, (720, 1280), dtype=np.uint8)
                uv = rng.integers(16, 241, (360, 640, 2), dtype=np.uint8)
                row[i] = ctx.frame(hip.FRAME_NV12, 1280, 720, [y, uv])
        r.register_shader("soften")
        r.register_image("logo", rng.integers(0, 256, (90, 160, 4), dtype=np.uint8))
        def second_scene(k):
            sc2 = synth.animated_grid_scene(bench.N_IN, k, 480, 270, 300, 2.5)
            sc2["children"].append({"type": "view", "width": 200, "height": 120, "right": 20, "bottom": 20, "children": [
                {"type": "image", "image_id": "logo"}, {"type": "text", "text": f"scene {k % 2}", "font_size": 28.0, "color": "#FFFF00FF"}]})
            return sc2
        second = second_scene
    sets = [r.make_frame_set({f"input_{i}": row[i] for i in range(bench.N_IN)}) for row in ring]
    sc = scenes()
    ns = 1_000_000_000 // 60
    proc = psutil.Process()
    frame = 0
    which = 0
    r.update_scene("out", bench.OUT_W, bench.OUT_H, sc[0])

    def picture():
        """the frames of the scenes at rest for input set 0 (rendered now, on whichever lane is next): every output's planes"""
        nonlocal frame
        n = r.render_packed(frame * ns, sets[0])
        frame += 1
        planes = []
        for k in range(n):
            planes += list(r.output(k).download())
        c = 0
        for p_ in planes:
            c = zlib.crc32(p_.tobytes(), c)
        return c, tuple(planes)

    # warm-up: both scenes seen, both lanes own their surfaces
    for k in range(2):
        r.update_scene("out", bench.OUT_W, bench.OUT_H, sc[k])
        if second:
            r.update_scene("second", 1920, 1080, second(k), hip.FRAME_NV12)
        for _ in range(120):
            r.render_packed(frame * ns, sets[frame % 4])
            frame += 1
    r.sync()
    which = 1
    first = {}
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = proc.memory_info().rss
    min_free, max_rss = free0, rss0
    updates = reregistrations = checks = 0
    mismatches = []
    rss_third = None
    t0 = time.perf_counter()
    frames0 = frame
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(400):
            r.render_packed(frame * ns, sets[frame % 4])
            frame += 1
        if updates % 5 == 4:  # an input goes away for a while (its tile shows nothing: populate_inputs) and comes back
            r.unregister_input("input_5")
            for _ in range(40):
                r.render_packed(frame * ns, sets[frame % 4])
                frame += 1
            r.register_input("input_5")
            reregistrations += 1
        # at rest by now (the transition took 18 frames): the picture of this scene for input set 0 is what it was the first time
        c, planes = picture()
        key = which
        if key not in first:
            first[key] = (c, planes)
        elif first[key][0] != c:
            import numpy as np
            mismatches.append({"frame": frame, "scene": key, "lane": frame % 2, "after_reregistration": updates % 5 == 4,
                               "differing_bytes": [int((a != b).sum()) for a, b in zip(planes, first[key][1])],
                               "max_abs_diff": int(max((np.abs(a.astype(int) - b.astype(int)).max() for a, b in zip(planes, first[key][1])), default=0))})
        checks += 1
        which ^= 1
        r.update_scene("out", bench.OUT_W, bench.OUT_H, sc[which])
        if second:
            r.update_scene("second", 1920, 1080, second(which), hip.FRAME_NV12)
        updates += 1
        r.sync()
        min_free = min(min_free, torch.cuda.mem_get_info()[0])
        if rss_third is None and time.perf_counter() - t0 > args.seconds / 3.0:
            rss_third = proc.memory_info().rss
        max_rss = max(max_rss, proc.memory_info().rss)
    r.sync()
    dt = time.perf_counter() - t0
    result = {"seconds": round(dt, 1), "frames": frame - frames0, "frames_per_s": round((frame - frames0) / dt, 1), "scene_updates": updates,
              "input_reregistrations": reregistrations, "picture_checks": checks, "distinct_pictures": len(first),
              "device_memory_growth_MB": round((free0 - min_free) / 2**20, 2), "host_rss_growth_MB": round((max_rss - rss0) / 2**20, 2),
              "host_rss_growth_last_two_thirds_MB": round((max_rss - (rss_third or rss0)) / 2**20, 2),
              "font_book": book is not None, "variety": bool(args.variety), "lanes": 3 if args.variety else 2, "outputs": 2 if second else 1, "picture_mismatches": len(mismatches), "first_mismatches": mismatches[:6]}
    print(json.dumps(result))
    assert result["device_memory_growth_MB"] < 64 and result["host_rss_growth_last_two_thirds_MB"] < 32 and not mismatches, result


if __name__ == "__main__":
    main()
