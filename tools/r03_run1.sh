#!/bin/bash
# round 3, first GPU call: the fused tests with the new kernel, the three-kernel A/B, occupancy knobs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -x -q > $O/r03_run1_fused.log 2>&1
tail -5 $O/r03_run1_fused.log
SMR_DEBUG_INGEST=1 timeout 600 python tools/ingest_ab.py 30 > $O/r03_run1_ab.log 2>&1
grep -v "^\[smr\]" $O/r03_run1_ab.log | grep "k_ingest_wave" | sort | uniq -c | head -5
grep "^{" $O/r03_run1_ab.log
for W in 2 3 4 5 6; do
  echo "wg_per_cu $W"; SMR_INGEST_WG_PER_CU=$W timeout 300 python tools/ingest_ab.py 30 --impls wave --contents bench 2>&1 | grep "^{" | cut -c1-200
done
timeout 600 python tools/ingest_ab.py 20 --4k --contents bench > $O/r03_run1_ab4k.log 2>&1
grep "^{" $O/r03_run1_ab4k.log
