#!/bin/bash
# ad-hoc PMC passes over tools/ingest_ab.py (wave kernel only): tools/r03_pmc.sh TAG "COUNTERS..." ["COUNTERS..." ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- python $R/tools/ingest_ab.py 20 --impls ${IMPL:-wave} --contents bench > $OUT/pmc$i.log 2>&1
done
cd $R
python tools/prof_summary.py $OUT 2>&1 | grep -v "^==" 
