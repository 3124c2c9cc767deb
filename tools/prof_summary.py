"""Summarises rocprofv3 CSV output (kernel stats + PMC passes) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for k in ("k_ingest_wave", "k_build_wave_weights", "k_yuv420_to_rgba", "k_yuv_to_rgba_batch", "k_ingest_resample", "k_compose_output", "k_classify_tiles", "k_apply_layouts", "k_build_weights", "k_resample_pass", "k_yuv_to_rgba",
              "k_rgba_to_y", "k_rgba_to_chroma", "k_blit_glyphs", "k_downsample"):
        if k in name:
            return k
    return name[:60]


for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-22s calls %6s  total %12s ns  avg %10s ns  min %10s  max %10s  %6s%%" % (
            short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("MinNs"),
            row.get("MaxNs"), row.get("Percentage")))

for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0)
            cnt[(k, row.get("Counter_Name"))] += 1
        print("== pmc:", os.path.relpath(f, out))
        for k, cs in acc.items():
            if k not in ("k_ingest_wave", "k_yuv420_to_rgba", "k_yuv_to_rgba_batch", "k_ingest_resample", "k_compose_output"):
                continue
            print("  " + k)
            for c, v in sorted(cs.items()):
                n = cnt[(k, c)]
                print("     %-28s per-dispatch avg %16.1f   (dispatches %d)" % (c, v / max(n, 1), n))

# ---- HBM traffic per launch for bench.py's roofline.traffic (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KB
#      at the L2's memory side, collected in separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes read -> x2.
import json

traffic = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(d) else []:
        acc, cnt = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(f)):
            c = row.get("Counter_Name")
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                k = short(row.get("Kernel_Name", ""))
                acc[(k, c)] += float(row.get("Counter_Value", 0) or 0)
                cnt[(k, c)] += 1
        for (k, c), v in acc.items():
            traffic.setdefault(k, {})[c] = v / cnt[(k, c)]
res = {}
for k, t in traffic.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        res[k] = {"fetch_kb_raw": round(t["FETCH_SIZE"], 1), "write_kb": round(t["WRITE_SIZE"], 1),
                  "hbm_bytes_per_launch": int(t["FETCH_SIZE"] * 1024 * 2 + t["WRITE_SIZE"] * 1024),
                  "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported"}
if res:
    json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("== traffic:", json.dumps(res))
