#!/bin/bash
# round 3: configs[4] (16 x 1080p -> 4K grid in transition: scales around 2) with and without the <8, 2> class of k_ingest_wave
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for name in "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for c in 4 2; do
    echo "== $name config $c: $(SMR_LIB=$PWD/$lib python bench.py --config $c --no-target --no-cpu-baseline --latency-frames 200 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernels'])")"
  done
done
