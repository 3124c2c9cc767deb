#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_11; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 400 --warmup 40 --latency-frames 100"
for a in "" "--config 3" "--config 1" "--config 4" "--ingest fused"; do
timeout 200 python bench.py $B $a 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()}, 'p50', r['latency_ms']['p50'])" | tee -a $O/bench.txt
done
bash tools/gpu_tests.sh r04_11
