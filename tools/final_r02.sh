#!/bin/bash
# Round-end check on the GPU box: the whole -m gpu suite, smoke(), the sharded path with one rank, a one-frame-in-flight kernel
# trace (a kernel's average there is its stand-alone duration: the figure roofline.avg_launch_us must agree with) and the bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/final_test.log 2>&1
grep -E "passed|failed|rror" $O/final_test.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --force-sharded --no-cpu-baseline --no-target --steps 50 --latency-frames 50 2>&1 | tail -1 | cut -c1-400
PROF_GROUPS=0 bash tools/prof.sh r02_one --inflight 1 --no-target > $O/prof_r02_one.log 2>&1
head -8 $O/prof_r02_one/summary.txt
python bench.py > $O/r02_bench_final.json 2> $O/r02_bench_final.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["frames_per_s_one_in_flight"], d["kernels"], d["latency_ms"], d["roofline"]["frac"], d["target"]["frames_per_s"], d["cpu_baseline"]["value"])
PY
