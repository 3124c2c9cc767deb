#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run8
mkdir -p $O
cd $R
bash tools/gpu_tests.sh r05_a
B="--no-cpu-baseline --no-target --latency-frames 200"
for n in 1 2 3; do
  timeout 200 python bench.py $B --inflight $n > $O/bench_inflight$n.json 2>/dev/null
done
timeout 200 python bench.py $B --config 4 > $O/bench_c4.json 2>/dev/null
timeout 200 python bench.py $B --config 1 > $O/bench_c1.json 2>/dev/null
hipcc --offload-arch=gfx950 -O3 tools/ubench/cvt_pk_u8.hip -o /tmp/cvt_pk_u8 2>/dev/null && /tmp/cvt_pk_u8 > $O/cvt_pk_u8.txt 2>&1; head -30 $O/cvt_pk_u8.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), r["value"], "fps; long", r.get("value_long",{}).get("frames_per_s"), "; 1 in flight", r["config"].get("frames_per_s_one_in_flight"), {k:v["avg_us"] for k,v in r.get("kernels",{}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
