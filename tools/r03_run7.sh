#!/bin/bash
# round 3: compositor occupancy A/B (libraries built by tools/variant.sh) on configs[4] and configs[2], one frame in flight
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for name in "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for c in 4 2; do
    echo "== $name config $c: $(SMR_LIB=$PWD/$lib python bench.py --config $c --no-target --no-cpu-baseline --latency-frames 100 --inflight 1 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernels'])")"
  done
done
