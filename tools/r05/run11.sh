#!/bin/bash
# converter residency cap (LDS pad) against co-residency with the other lane's kernels: pipelined bench, laboratory build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run11
mkdir -p $O
cd $R
export SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so
B="--no-cpu-baseline --no-target --latency-frames 100 --long-seconds 3"
for cfg in "4:0" "4:256" "5:256" "6:256" "3:256" "4:0" "4:256" "6:0"; do
  k=${cfg%%:*}; pad=${cfg#*:}
  SMR_CONVERT_WG_PER_CU=$k SMR_CONVERT_LDS_PAD=$pad timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k $k pad $pad:', r['value'], 'long', r['value_long']['frames_per_s'], 'one', r['config']['frames_per_s_one_in_flight'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
