#!/bin/bash
# round 4, GPU call 1: the exact-converter default — tests, converter / resampler A/B, kernel trace.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
tools/ubench/build/cvt_pk_u8 > $O/cvt_pk_u8.txt 2>&1
B="--no-cpu-baseline --no-target --steps 300 --warmup 30 --latency-frames 200"
for v in "auto:" "fused:--ingest fused" "block4x2:--convert block4x2" "general:--convert general"; do
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
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.rg3.so timeout 300 python bench.py $B > $O/bench_rg3.json 2> $O/bench_rg3.err
python - <<PY
import json
r=json.loads(open("$O/bench_rg3.json").read().strip().splitlines()[-1])
print("rg3", r["value"], "fps; one in flight", r["config"]["frames_per_s_one_in_flight"], {k:v["avg_us"] for k,v in r["kernels"].items()})
PY
timeout 300 python bench.py --config 3 $B > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
r=json.loads(open("$O/bench_c3.json").read().strip().splitlines()[-1])
print("configs3", r["value"], "fps; one in flight", r["config"]["frames_per_s_one_in_flight"], {k:v["avg_us"] for k,v in r["kernels"].items()})
PY
PROF_GROUPS=0 bash tools/prof.sh r04_1 --inflight 1 --no-target > $O/prof.log 2>&1
cp gpurun_out/prof_r04_1/stats/*kernel_stats.csv $O/ 2>/dev/null
head -12 $O/*kernel_stats.csv | cut -c1-200
