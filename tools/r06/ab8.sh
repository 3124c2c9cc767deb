#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/ab9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_renderer.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR| passed| failed" | tail -5
for n in product prev; do
  lib=smelter_amd/variants/libsmr_hip.$n.so; [ $n = product ] && lib=smelter_amd/libsmr_hip.so
  for rep in 1 2; do
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_${n}_$rep.json 2> $O/bench_${n}_$rep.err
  done
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 > $O/bench_${n}_if1.json 2> $O/bench_${n}_if1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab9/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
bash tools/r06/tl.sh | head -12
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target --no-long | python -c "import sys,json; print('driver-like', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; done
