#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_5; mkdir -p $O
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.timing.so timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --steps 20 --warmup 5 --latency-frames 5 --inflight 1 2>&1 | grep "kcycles" | tail -3 | tee $O/timing.txt
