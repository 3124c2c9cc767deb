#!/bin/bash
# Round-2 evidence run on the GPU box (one gpurun call): A/B table, issue rates, kernel-trace stats + PMC passes of the judged
# config, HBM traffic with direct output on, kernel-trace stats of the north-star target config, the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
python tools/mfma_ab.py 50 > $O/r02_mfma_ab.log 2>&1
tools/ubench/build/valu_rate > $O/r02_valu_rate.txt 2>&1
bash tools/prof.sh r02 > $O/prof_r02.log 2>&1
cd /tmp
B="python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-target --latency-frames 5"
for G in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/prof_r02_direct/$G -o p -- $B --direct-output > $O/prof_r02_direct_$G.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_configs3/stats -o s -- $B --config 3 > $O/prof_r02_configs3.log 2>&1
for G in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/prof_r02_configs3/$G -o p -- $B --config 3 > $O/prof_r02_configs3_$G.log 2>&1
done
cd $R
python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
tail -c 600 $O/r02_bench.json
