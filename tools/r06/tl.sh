#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/tl; mkdir -p $O
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/trace -o t -- python $R/bench.py --steps 400 --warmup 30 --no-cpu-baseline --no-target --no-long --latency-frames 5 > $R/$O/trace.log 2>&1)
python tools/timeline.py $(find gpurun_out/tl/trace -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline_two_lanes.txt 2>&1
find gpurun_out/tl/trace -name "*kernel_trace.csv" -delete
head -40 $O/timeline_two_lanes.txt
