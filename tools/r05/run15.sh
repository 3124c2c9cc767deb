#!/bin/bash
# resident workgroups of the resampler capped (room for the other lane's kernels): pipelined bench, laboratory build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so
B="--no-cpu-baseline --no-target --latency-frames 100 --long-seconds 3"
for w in 0 4 3 0 4 5; do
  SMR_INGEST_WG_PER_CU=$w timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ingest wg/cu $w:', r['value'], 'long', r['value_long']['frames_per_s'], 'one', r['config']['frames_per_s_one_in_flight'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
