#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/prime; mkdir -p $O
for ps in 0 0.25 1.0; do for i in 1 2 3; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target --no-long --prime-seconds $ps > $O/d_${ps}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prime/d_*.json')):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], r['value'])
PY
