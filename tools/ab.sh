#!/bin/bash
# On the GPU box: bench every variant library built by tools/variant.sh (and the normal build as "base").
#   tools/ab.sh [--config N] NAME...
cd "$(dirname "$0")/.."
cfg=2
if [ "$1" = "--config" ]; then cfg=$2; shift 2; fi
for name in base "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  SMR_LIB=$PWD/$lib timeout 200 python bench.py --config $cfg --no-cpu-baseline --steps 300 --warmup 30 --latency-frames 20 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$name', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
