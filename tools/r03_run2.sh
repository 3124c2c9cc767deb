#!/bin/bash
# round 3, GPU call 2: variants of k_ingest_wave (registers / occupancy / weight prefetch / waves per workgroup)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
for name in base "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for W in 0 ; do
    echo "== $name"; SMR_LIB=$PWD/$lib SMR_DEBUG_INGEST=1 timeout 300 python tools/ingest_ab.py 40 --impls wave --contents bench 2>&1 | grep "^{" | cut -c1-120
  done
done
