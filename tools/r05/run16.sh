#!/bin/bash
# the converter's banded (XCD-aware) partition: parity, kernel trace, FETCH_SIZE
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run16
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
cd /tmp
BENCH="python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-long --no-target --latency-frames 5 --inflight 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $BENCH > $O/stats.log 2>&1
grep -h "k_yuv420_to_rgba" $(find $O/stats -name "*kernel_stats.csv") | awk -F, '{print "converter: calls", $(NF-6), "avg ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}'
for G in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/pmc_$G -o p -- $BENCH > $O/pmc_$G.log 2>&1
done
cd $R
python tools/traffic_json.py $O | python -c "
import json,sys; t=json.load(sys.stdin)
for k in ('k_yuv420_to_rgba','k_ingest_wave','k_compose_output'): print(k, t[k])"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
B="--no-cpu-baseline --no-target --latency-frames 100 --long-seconds 3"
for i in 1 2; do timeout 200 python bench.py $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', r['value'], 'long', r['value_long']['frames_per_s'], 'one', r['config']['frames_per_s_one_in_flight'], {k:v['avg_us'] for k,v in r['kernels'].items()})"; done
