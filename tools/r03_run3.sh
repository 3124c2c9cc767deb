#!/bin/bash
# round 3: the single-tile experiment (SMR_WAVE_ONE_TILE) and the default kernel against resident workgroups per CU (2 waves each)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
for name in "$@"; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for W in 2 3 4 5 6 8; do
    echo "== $name wg/cu $W"; SMR_INGEST_WG_PER_CU=$W SMR_LIB=$PWD/$lib SMR_DEBUG_INGEST=1 timeout 300 python tools/ingest_ab.py 40 --impls wave --contents bench 2>&1 | grep "^{" | cut -c1-70
  done
done
