#!/bin/bash
# the converter's solo duration in the kernel trace (no stamps, one frame in flight) against resident workgroups per CU / LDS cap
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run13
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
export SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so
for cfg in "4:0" "5:0" "6:0" "6:256" "4:256" "3:0"; do
  k=${cfg%%:*}; pad=${cfg#*:}
  SMR_CONVERT_WG_PER_CU=$k SMR_CONVERT_LDS_PAD=$pad timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k${k}_p$pad -o s -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-long --no-target --latency-frames 5 --inflight 1 > $O/k${k}_p$pad.log 2>&1
  echo "k $k pad $pad: $(grep -h 'k_yuv420_to_rgba' $O/k${k}_p$pad/*/*kernel_stats.csv $O/k${k}_p$pad/*kernel_stats.csv 2>/dev/null | head -1 | awk -F, '{print "calls", $(NF-6), "avg ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}')"
done
find $O -name "*kernel_trace.csv" -delete
