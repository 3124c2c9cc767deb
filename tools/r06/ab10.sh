#!/bin/bash
# co-residency experiment: fewer registers in the kernels that run beside the other lane's resampler
cd "$(dirname "$0")/../.."
O=gpurun_out/ab12; mkdir -p $O
for n in base3 nf5; do
  lib=smelter_amd/variants/libsmr_hip.$n.so
  for rep in 1 2; do
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_${n}_$rep.json 2> $O/bench_${n}_$rep.err
  done
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 > $O/bench_${n}_if1.json 2> $O/bench_${n}_if1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab12/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
