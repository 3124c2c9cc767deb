#!/bin/bash
# round 6, first contact: the fast output conversion + the cheaper encode on hardware — parity tests first, then the bench A/B (default | direct output)
cd "$(dirname "$0")/../.."
O=gpurun_out/ab1; mkdir -p $O
python -c "from smelter_amd import build; print(build.kernels_sha256())" > $O/lib_identity.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_renderer.py -q -m gpu -x 2>&1 | tail -n 8 > $O/pytest.txt
for v in "default:" "direct:--direct-output"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 $f > $O/bench_$n.json 2> $O/bench_$n.err
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 $f > $O/bench_${n}_if1.json 2> $O/bench_${n}_if1.err
done
cat $O/pytest.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab1/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
