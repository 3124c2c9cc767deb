"""Wave-instruction counts per frame and the instruction-issue floors they imply, from a tools/prof.sh summary (the PMC passes SQ_INSTS_VALU /
SQ_INSTS_LDS / SQ_INSTS_MFMA of the bench command).   python tools/issue_json.py profiles/r05_rocprofv3_summary.txt > profiles/r05_issue.json

The vector ALU and the matrix pipe of a SIMD are SEPARATE pipes that run side by side (MI355X_MICROARCH.md, "Wave scheduling": "MFMA and VALU
pipes are separate ... both ~max, not sum"), so the floor of a frame is the LARGER of the two pipes' times, not their sum (round 4 added
them).  Three prices are reported side by side — none of them is "the" floor, they bracket it:
  guide      every vector instruction at the guide's 2 cycles per wave64 instruction, 2.4 GHz (0.83 ns); an MFMA 16x16x32 f16 at 17 cycles per SIMD
  measured   this device's measured issue rates (profiles/r02_valu_rate.txt): 1.1 ns for fp32 mul / add / fma, 1.8 ns for conversions, permutes,
             SDWA and three-operand integer ops, mixed 55 / 45 as the three kernels' ISA is; MFMA 7.4 ns
  stamped    what the converter's wave stamps show a saturated SIMD delivers on this code (profiles/r05_convert_waves.txt): 3.9 cycles per vector
             instruction at the 2.05 GHz the chip holds under this load = 1.9 ns"""
import json
import os
import re
import sys

KERNELS = ["k_yuv420_to_rgba", "k_ingest_wave", "k_compose_output"]
SIMDS = 256 * 4
PRICES = {  # ns per wave-instruction per SIMD: (vector, mfma)
    "guide": (2.0 / 2.4, 17.0 / 2.4),
    "measured": (0.55 * 1.1 + 0.45 * 1.8, 7.4),
    "stamped": (3.9 / 2.05, 7.4),
}


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
    floors = {}
    for name, (pv, pm) in PRICES.items():
        tv, tm = valu * pv / SIMDS / 1000.0, mfma * pm / SIMDS / 1000.0
        floors[name] = {"vector_us": round(tv, 1), "matrix_us": round(tm, 1), "floor_us_pipes_overlapped": round(max(tv, tm), 1),
                        "upper_us_pipes_serialised": round(tv + tm, 1), "ns_per_vector_instruction": round(pv, 3), "ns_per_mfma": round(pm, 2)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.traffic_json import lib_identity
    print(json.dumps({
        "source": f"{path} (rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_MFMA, one frame in flight, configs[2])",
        "_identity": lib_identity(),
        "per_kernel": {k: out[k] for k in KERNELS},
        "per_frame": {"valu_wave_instructions": valu, "mfma_wave_instructions": mfma, "lds_wave_instructions": lds},
        "simds": SIMDS,
        "floors_us_per_frame": floors,
        "reading": "the vector and matrix pipes overlap: a frame cannot be shorter than floor_us_pipes_overlapped at the given price; "
                   "'guide' is the hardware's peak issue rate, 'stamped' what a saturated SIMD has been seen to deliver on this instruction mix",
        "issue_floor_us_per_frame": floors["guide"]["floor_us_pipes_overlapped"]}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
