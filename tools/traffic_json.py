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


def lib_identity():
    """Which library's kernels the counters belong to: bench.py quotes a traffic file only for the library it is timing (smelter_amd/build.py)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from smelter_amd import build as B
    lib = os.environ.get("SMR_LIB") or B.LIB
    return {"lib_kernels_sha256": B.kernels_sha256(lib), "lib": os.path.relpath(lib, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}


if __name__ == "__main__":
    out = collect(sys.argv[1])
    out["_identity"] = lib_identity()
    print(json.dumps(out, indent=1))
