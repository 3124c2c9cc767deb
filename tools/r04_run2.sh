#!/bin/bash
# round 4, GPU call 2: planar node textures + direct output on the default route.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
B="--no-cpu-baseline --no-target --steps 300 --warmup 30 --latency-frames 200"
run() { n=$1; shift; env "$@" > /dev/null 2>&1; }
for v in "auto:" "direct:--direct-output" "fused:--ingest fused" "inflight1:--inflight 1" "inflight2:--inflight 2" "direct_inflight2:--direct-output --inflight 2"; do
  n=${v%%:*}; a=${v#*:}
  timeout 300 python bench.py $B $a > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
    print("$n", r["value"], "fps; long", r["value_long"]["frames_per_s"], "; one in flight", r["config"]["frames_per_s_one_in_flight"], {k:v["avg_us"] for k,v in r["kernels"].items()}, "p50", r["latency_ms"]["p50"])
except Exception as e:
    print("$n failed", e)
PY
done
SMR_PLANAR_NODES=0 timeout 300 python bench.py $B > $O/bench_rgba8nodes.json 2> $O/bench_rgba8nodes.err
python - <<PY
import json
r=json.loads(open("$O/bench_rgba8nodes.json").read().strip().splitlines()[-1])
print("rgba8 nodes", r["value"], "fps; one in flight", r["config"]["frames_per_s_one_in_flight"], {k:v["avg_us"] for k,v in r["kernels"].items()})
PY
for c in 3 1 4; do
timeout 300 python bench.py --config $c $B > $O/bench_c$c.json 2> $O/bench_c$c.err
python - <<PY
import json
r=json.loads(open("$O/bench_c$c.json").read().strip().splitlines()[-1])
print("configs$c", r["value"], "fps; one in flight", r["config"]["frames_per_s_one_in_flight"], {k:v["avg_us"] for k,v in r["kernels"].items()})
PY
done
PROF_GROUPS=0 bash tools/prof.sh r04_2 --inflight 1 --no-target > $O/prof.log 2>&1
cp gpurun_out/prof_r04_2/stats/*kernel_stats.csv $O/ 2>/dev/null
PROF_GROUPS=0 bash tools/prof.sh r04_2d --inflight 1 --no-target --direct-output > $O/prof_d.log 2>&1
cp gpurun_out/prof_r04_2d/stats/s_kernel_stats.csv $O/s_kernel_stats_direct.csv 2>/dev/null
head -5 $O/s_kernel_stats.csv $O/s_kernel_stats_direct.csv | cut -c1-150
