#!/bin/bash
# the -m gpu suite on the product build, smoke(), the driver's bench command (what the round-end driver runs), in one GPU call
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/verify
python -c "from smelter_amd import build; print(build.kernels_sha256())" > gpurun_out/verify/lib_identity.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -n 15 > gpurun_out/verify/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/verify/smoke.txt 2>&1
s=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/verify/bench_driver_like.json 2> gpurun_out/verify/bench.err; e=$(date +%s)
echo "driver command took $((e - s)) s" >> gpurun_out/verify/bench.err
tail -n 4 gpurun_out/verify/pytest_gpu.txt; cat gpurun_out/verify/smoke.txt | tail -n 2; python -c "
import json; r=json.loads(open('gpurun_out/verify/bench_driver_like.json').read().strip().splitlines()[-1]); print(r['value'], r.get('value_long'), r['roofline'].get('traffic_stale'), r['roofline']['frac'])"; tail -n 1 gpurun_out/verify/bench.err
