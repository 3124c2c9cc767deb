#!/bin/bash
# Everything the round's evidence needs from the GPU box, in one call: the -m gpu suite, smoke(), the sharded path with one rank, the
# host-thread rate, a one-frame-in-flight kernel trace, then tools/prof_r02.sh (A/B table, issue rates, three-in-flight trace + PMC
# passes, traffic with direct output, configs[3] trace + traffic, the default bench line).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/final_test.log 2>&1
grep -E "passed|failed|rror" $O/final_test.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --force-sharded --no-cpu-baseline --no-target --steps 50 --latency-frames 50 2>&1 | tail -1 | cut -c1-200
python tools/host_rate.py 2>&1 | tail -1 | tee $O/r02_host_rate.txt
PROF_GROUPS=0 bash tools/prof.sh r02_one --inflight 1 --no-target > $O/prof_r02_one.log 2>&1
head -6 $O/prof_r02_one/summary.txt
bash tools/prof_r02.sh
