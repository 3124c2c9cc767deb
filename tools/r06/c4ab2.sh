#!/bin/bash
# two laboratory builds on configs[4] frozen in motion (stage timers; SMR_ABLATE 8192 = no sampled tiles, 4096 = no composited tiles), then the benches
cd "$(dirname "$0")/../.."
for v in "$@"; do
  L=$PWD/smelter_amd/variants/libsmr_hip.$v.so
  echo "== $v"
  for ab in 0 8192 4096; do SMR_LIB=$L SMR_ABLATE=$ab timeout 300 python tools/r06/c4probe.py 200 0.5 2>&1 | tail -1; done
done
bash tools/r06/abe.sh 4 2 "$@"
bash tools/r06/abe.sh 2 2 "$@"
