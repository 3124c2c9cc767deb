#!/bin/bash
# is the pipelined rate sensitive to what the Infinity Cache holds?  input ring of 1 / 2 / 12 frame sets (25 MB each)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
B="--no-cpu-baseline --no-target --latency-frames 100 --long-seconds 3"
for ring in 1 2 4 12 1 12; do
  SMR_BENCH_RING=$ring timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring $ring:', r['value'], 'long', r['value_long']['frames_per_s'], 'one', r['config']['frames_per_s_one_in_flight'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
