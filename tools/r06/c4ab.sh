#!/bin/bash
# block compositor (lab0) against the pixel-per-sweep one (labold): configs[4] frozen in motion, then the benches
cd "$(dirname "$0")/../.."
O=gpurun_out/c4ab; mkdir -p $O
for v in lab0 labold; do
  L=$PWD/smelter_amd/variants/libsmr_hip.$v.so
  echo "== $v"
  for ab in 0 8192 4096; do SMR_LIB=$L SMR_ABLATE=$ab timeout 300 python tools/r06/c4probe.py 200 0.5 2>&1 | tail -1; done
  for c in 4 2; do
    SMR_LIB=$L timeout 600 python bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 2 > $O/bench_${v}_c$c.json 2> $O/bench_${v}_c$c.err
  done
done
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_reference_scenes.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | tail -5
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c4ab/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), r['config'].get('frames_per_s_one_in_flight'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
