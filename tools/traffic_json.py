"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md HBM section:
KB at the L2's memory side; on gfx950 FETCH_SIZE reports half of the bytes read -> x2).
python tools/traffic_json.py <dir with the two passes anywhere below it>  -> JSON on stdout"""
import csv
import glob
import json
import os
import sys

KERNELS = ("k_ingest_wave", "k_yuv420_to_rgba", "k_yuv_to_rgba_batch", "k_ingest_resample", "k_compose_output", "k_classify_tiles", "k_apply_layouts", "k_build_weights",
           "k_resample_pass", "k_yuv_to_rgba", "k_rgba_to_y", "k_rgba_to_chroma", "k_blit_glyphs", "k_downsample", "k_gauss_axis")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return name[:60]


def collect(root):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r.get("Counter_Name")
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = short(r["Kernel_Name"])
            tot[(k, c)] = tot.get((k, c), 0.0) + float(r["Counter_Value"])
            cnt[(k, c)] = cnt.get((k, c), 0) + 1
    res = {}
    for (k, c), v in tot.items():
        res.setdefault(k, {})[c] = v / cnt[(k, c)]
    out = {}
    for k, t in res.items():
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            out[k] = {"fetch_kb_raw": round(t["FETCH_SIZE"], 1), "write_kb": round(t["WRITE_SIZE"], 1),
                      "hbm_bytes_per_launch": int(t["FETCH_SIZE"] * 1024 * 2 + t["WRITE_SIZE"] * 1024),
                      "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported"}
    return out


if __name__ == "__main__":
    print(json.dumps(collect(sys.argv[1]), indent=1))
