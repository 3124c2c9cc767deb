#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_8; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 400 --warmup 40 --latency-frames 50"
for pad in 0 30000 48000 64000; do
 for inf in 2 3 4; do
  SMR_CONVERT_LDS_PAD=$pad timeout 200 python bench.py $B --inflight $inf 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $pad inflight $inf', r['value'], 'fps', {k:v['avg_us'] for k,v in r['kernels'].items()})" | tee -a $O/bench.txt
 done
done
