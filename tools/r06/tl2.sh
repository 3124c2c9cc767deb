#!/bin/bash
# two-lane timelines of laboratory builds:  tools/r06/tl2.sh NAME [NAME ...]
cd "$(dirname "$0")/../.."
R=$PWD
for v in "$@"; do
O=gpurun_out/tl2_$v; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && SMR_LIB=$R/smelter_amd/variants/libsmr_hip.$v.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/bench.py --steps 400 --warmup 30 --no-cpu-baseline --no-target --no-long --latency-frames 5 > $R/$O/trace.log 2>&1)
python tools/timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline.txt 2>&1
find $O/trace -name "*kernel_trace.csv" -delete
echo "== $v"; head -22 $O/timeline.txt
done
