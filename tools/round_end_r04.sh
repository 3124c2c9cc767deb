#!/bin/bash
# Round 4's evidence from the GPU box in one call: the -m gpu suite, smoke(), the default bench line (CPU baseline + target block),
# the other configs and routes, the sharded code path with one rank, a one-frame-in-flight kernel trace + PMC passes of configs[2],
# kernel trace + traffic passes of configs[3].  Results under gpurun_out/r04_end/ (copied into profiles/ by hand).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_end
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_tests.sh r04_end
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target > $O/bench_driver_like.json 2>/dev/null
for a in "c1:--config 1" "c3:--config 3" "c4:--config 4" "fused:--ingest fused" "valu:--ingest valu" "inflight1:--inflight 1" "inflight3:--inflight 3"; do
  n=${a%%:*}; f=${a#*:}
  timeout 300 python bench.py --no-cpu-baseline --no-target $f > $O/bench_$n.json 2>/dev/null
done
timeout 600 python tools/ingest_ab.py 30 > $O/ingest_ab.log 2>&1; cp gpurun_out/r04_ingest_ab.json $O/ 2>/dev/null
timeout 300 python bench.py --force-sharded --no-cpu-baseline --no-target --steps 100 --latency-frames 50 > $O/bench_sharded_1rank.json 2>$O/bench_sharded.err
bash tools/prof.sh r04_final --inflight 1 --no-target > $O/prof_final.log 2>&1
PROF_GROUPS=0 bash tools/prof.sh r04_c3 --config 3 --inflight 1 > $O/prof_c3.log 2>&1
for G in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04_c3/pmc_$G -o p -- python $R/bench.py --config 3 --steps 60 --warmup 10 --no-cpu-baseline --no-long --latency-frames 5 --inflight 1 > $O/prof_c3_$G.log 2>&1)
done
python tools/traffic_json.py gpurun_out/prof_r04_c3 > $O/traffic_configs3.json 2>/dev/null
find gpurun_out/prof_r04_c3 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r04_c3 -name "*counter_collection.csv" -delete
PROF_GROUPS=0 bash tools/prof.sh r04_c4 --config 4 --inflight 1 > $O/prof_c4.log 2>&1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), r["value"], "fps; long", r.get("value_long",{}).get("frames_per_s"), "; 1 in flight", r["config"].get("frames_per_s_one_in_flight"), {k:v["avg_us"] for k,v in r.get("kernels",{}).items()}, "p50", r.get("latency_ms",{}).get("p50"))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
