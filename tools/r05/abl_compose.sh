#!/bin/bash
# compositor ablations (laboratory build, SMR_ABLATE bits 8..: 256 dispatch only | 512 classify only | 2048 base layer only | 4096 copy tiles only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so
for a in 0 256 4096 2048 0; do
  SMR_ABLATE=$a timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --inflight 1 --steps 200 --latency-frames 5 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $a:', r['value'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
