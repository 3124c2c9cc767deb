#!/bin/bash
# the whole -m gpu suite on the product build (summary lines), then the resampler's phase timing (laboratory timing build, one frame in flight)
cd "$(dirname "$0")/../.."
O=gpurun_out/full; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"
grep -E "^FAILED|^ERROR| passed| failed|error" $O/pytest.log | tail -8
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.wtime.so timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-target --no-long --inflight 1 > $O/bench_wtime.json 2> $O/bench_wtime.err
grep "kcycles" $O/bench_wtime.err | tail -3
