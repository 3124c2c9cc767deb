#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch use as hipcc reports it (-Rpass-analysis=kernel-resource-usage), one row per kernel.
usage: python tools/kernel_resources.py smelter_amd/csrc/smr_fused.hip [more .hip files] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smelter_amd import build as B

def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n

def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--"); extra = args[i + 1:]; args = args[:i]
    for src in args:
        cmd = [B.HIPCC] + B.FLAGS + extra + ["-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        rows, cur = [], None
        for line in err.splitlines():
            m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark: (.*) \[-Rpass", line)
            if not m: continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = {"name": demangle(t.split(":", 1)[1].strip())}; rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        print(f"# {src}")
        print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr':>5} {'occ':>4} {'LDS':>7}  kernel")
        for r in rows:
            nm = re.sub(r"\(anonymous namespace\)::", "", r["name"])
            nm = re.sub(r"\(.*$", "", nm)
            print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs', r.get('SGPRs','?')):>5} {r.get('ScratchSize [bytes/lane]','?'):>5} "
                  f"{r.get('Occupancy [waves/SIMD]','?'):>4} {r.get('LDS Size [bytes/block]','?'):>7}  {nm}")

if __name__ == "__main__":
    main()
