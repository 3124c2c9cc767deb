"""Wave-instruction counts per frame and the instruction-issue floor they imply, from a tools/prof.sh summary (the PMC passes SQ_INSTS_VALU / SQ_INSTS_LDS /
SQ_INSTS_MFMA of the bench command): how profiles/r04_issue.json is made.   python tools/issue_json.py profiles/r04_rocprofv3_summary_v3.txt > profiles/r04_issue.json"""
import json
import re
import sys

KERNELS = ["k_yuv420_to_rgba", "k_ingest_wave", "k_compose_output"]
FULL, SLOW, MFMA = 1.1, 1.8, 7.4  # ns per wave-instruction per SIMD, measured on this device (profiles/r02_valu_rate.txt, profiles/r03_valu_occ.txt)
SIMDS = 256 * 4


def main(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^  (k_\w+)\s*$", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s+(SQ_INSTS_VALU|SQ_INSTS_LDS|SQ_INSTS_MFMA|SQ_INSTS_SALU|SQ_WAVES)\s+per-dispatch avg\s+([\d.]+)", line)
        if m and cur:
            out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    valu = sum(out[k]["SQ_INSTS_VALU"] for k in KERNELS)
    mfma = sum(out[k].get("SQ_INSTS_MFMA", 0.0) for k in KERNELS)
    lds = sum(out[k]["SQ_INSTS_LDS"] for k in KERNELS)
    floor_us = (valu * (0.55 * FULL + 0.45 * SLOW) + mfma * MFMA) / SIMDS / 1000.0
    print(json.dumps({
        "source": f"{path} (rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_MFMA, one frame in flight, configs[2])",
        "per_kernel": {k: out[k] for k in KERNELS},
        "per_frame": {"valu_wave_instructions": valu, "mfma_wave_instructions": mfma, "lds_wave_instructions": lds},
        "simds": SIMDS,
        "issue_rates_ns": {"fp32_mul_add_fma": FULL, "conversions_permutes_shift_ors": SLOW, "mfma_16x16x32_f16": MFMA,
                           "assumed_mix": "55 % full rate / 45 % slow (ISA of the three kernels)", "source": "profiles/r02_valu_rate.txt, profiles/r03_valu_occ.txt"},
        "issue_floor_us_per_frame": round(floor_us, 1)}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
