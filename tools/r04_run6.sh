#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_6; mkdir -p $O
B="--no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 100"
for a in "" "--ingest fused" "--config 3" "--config 1" "--config 4"; do
timeout 200 python bench.py $B $a 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})" | tee -a $O/bench.txt
done
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
