#!/bin/bash
# configs[4] (animated grid + blur layer): every kernel on the timeline, two frames in flight and one
cd "$(dirname "$0")/../.."
O=gpurun_out/tl4; mkdir -p $O
R=$PWD
for fl in 2 1; do
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace$fl -o t -- python $R/bench.py --config 4 --inflight $fl --steps 400 --warmup 30 --no-cpu-baseline --no-target --no-long --latency-frames 5 > $R/$O/trace$fl.log 2>&1)
TL_ALL=1 TL_SAMPLE=140 python tools/timeline.py $(find $O/trace$fl -name "*kernel_trace.csv" | head -1) 0.3 > $O/timeline_c4_inflight$fl.txt 2>&1
find $O/trace$fl -name "*kernel_trace.csv" -delete
done
head -60 $O/timeline_c4_inflight2.txt
