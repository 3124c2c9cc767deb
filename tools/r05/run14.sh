#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run14
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_renderer.py tests/test_gpu_fused.py tests/test_gpu_reference_scenes.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="--no-cpu-baseline --no-target --latency-frames 100 --long-seconds 3"
for c in 4 4 2 2; do
timeout 200 python bench.py $B --config $c 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c:', r['value'], 'long', r['value_long']['frames_per_s'], 'one', r['config']['frames_per_s_one_in_flight'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
