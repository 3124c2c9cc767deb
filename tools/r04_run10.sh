#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_10; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 200 --warmup 20 --latency-frames 20 --inflight 1"
for name in base abl1 abl2 abl8 abl16 abl9 abl11; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  SMR_LIB=$PWD/$lib timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', {k:v['avg_us'] for k,v in r['kernels'].items()})" | tee -a $O/abl.txt
done
