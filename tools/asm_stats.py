#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file (hipcc -S --offload-device-only): total, vector ALU, matrix, LDS, memory,
scalar — per kernel whose mangled name contains one of the given substrings.  A quick before / after for instruction-count work
(the dynamic counts come from the counters: tools/issue_json.py).
usage: python tools/asm_stats.py file.s substr [substr ...]"""
import re, sys
from collections import Counter

def kernels(path):
    txt = open(path).read()
    for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
        yield m.group(1), m.group(2)

def mix(body):
    ins = []
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith((';', '.')) or l.endswith(':'):
            continue
        ins.append(l.split()[0])
    c = Counter(ins)
    g = lambda pred: sum(n for k, n in c.items() if pred(k))
    return {"total": len(ins), "valu": g(lambda k: k.startswith('v_') and not k.startswith('v_mfma')), "mfma": g(lambda k: k.startswith('v_mfma')),
            "lds": g(lambda k: k.startswith('ds_')), "vmem": g(lambda k: k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))),
            "salu": g(lambda k: k.startswith('s_'))}, c

if __name__ == "__main__":
    path, subs = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if subs and not any(s in name for s in subs):
            continue
        m, c = mix(body)
        print(name[:110])
        print("   ", m, "top:", ", ".join(f"{k} {n}" for k, n in c.most_common(8)))
