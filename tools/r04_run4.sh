#!/bin/bash
# round 4, GPU call 4: resampler variants on the default route (B band in registers, staggered waves)
cd "$(dirname "$0")/.."
O=gpurun_out/r04_4; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 100"
for name in base bregs stag40 stag80 bregs_stag40; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  SMR_LIB=$PWD/$lib timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})" | tee -a $O/variants.txt
done
