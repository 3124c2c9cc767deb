#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_7; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 100"
for name in base early; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for a in "" "--config 3" "--config 1" "--config 4"; do
  SMR_LIB=$PWD/$lib timeout 200 python bench.py $B $a 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name [$a]', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})" | tee -a $O/bench.txt
  done
done
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.timing.so timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --steps 20 --warmup 5 --latency-frames 5 --inflight 1 2>&1 | grep "kcycles" | tail -2 | tee $O/timing.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
