"""A soak run of the renderer on the benchmark scene: minutes of frames on two lanes with everything a live host does in between —
scene updates that start transitions (tiles reordered by id, an input leaving and coming back), inputs unregistered and registered
again, text drawn by the library's font book at every update — while watching what must not move:
  * device memory (hipMemGetInfo through torch) over the whole run and the host's resident set over its last two thirds (the first read-backs
    grow the runtime's staging pool once: ~200 MB over the first few hundred downloads, nothing after — measured with frames only, updates only
    and downloads only): no growth,
  * the picture: the frame of a scene at rest, for the same input set, has the same checksum every time it comes round,
  * the error channel: no call fails.
python tools/soak.py [--seconds 150]          prints one JSON line"""
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
    args = ap.parse_args()
    import psutil
    import torch
    ctx = hip.Context(0)
    r = Renderer(ctx, stream_fallback_timeout_s=1e9, lanes=[hip.Context(0)])  # (the ring's frames carry pts 0: never stale, however long the run)
    book = bench.font_book()
    if book is not None:
        r.set_fontbook(book)
    for i in range(bench.N_IN):
        r.register_input(f"input_{i}")
    ring = bench.make_inputs(ctx, hip, 4, list(range(bench.N_IN)))
    sets = [r.make_frame_set({f"input_{i}": row[i] for i in range(bench.N_IN)}) for row in ring]
    sc = scenes()
    ns = 1_000_000_000 // 60
    proc = psutil.Process()
    frame = 0
    which = 0
    r.update_scene("out", bench.OUT_W, bench.OUT_H, sc[0])

    def picture():
        """the frame of a scene at rest for input set 0 (rendered now, on whichever lane is next)"""
        nonlocal frame
        r.render_packed(frame * ns, sets[0])
        frame += 1
        out = r.output(0)
        y, u, v = out.download()
        return zlib.crc32(v.tobytes(), zlib.crc32(u.tobytes(), zlib.crc32(y.tobytes()))), (y, u, v)

    # warm-up: both scenes seen, both lanes own their surfaces
    for k in range(2):
        r.update_scene("out", bench.OUT_W, bench.OUT_H, sc[k])
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
              "font_book": book is not None, "picture_mismatches": len(mismatches), "first_mismatches": mismatches[:6]}
    print(json.dumps(result))
    assert result["device_memory_growth_MB"] < 64 and result["host_rss_growth_last_two_thirds_MB"] < 32 and not mismatches, result


if __name__ == "__main__":
    main()
