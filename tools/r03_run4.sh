#!/bin/bash
# round 3: compositor A/B on configs[4] (and configs[2]): libraries x SMR_COMPOSE_SLICES
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for name in "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for S in 8 4; do
    for c in 4 2; do
      echo "== $name slices $S config $c: $(SMR_COMPOSE_SLICES=$S SMR_LIB=$PWD/$lib python bench.py --config $c --no-target --no-cpu-baseline --latency-frames 200 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernels'])")"
    done
  done
done
