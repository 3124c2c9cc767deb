#!/bin/bash
# the driver's command (20 steps) three times + the capacity mode
cd "$(dirname "$0")/../.."
O=gpurun_out/cap; mkdir -p $O
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target --long-seconds 2 > $O/driver_like_$i.json 2> $O/driver_like_$i.err; done
timeout 900 python bench.py --mode capacity > $O/capacity.json 2> $O/capacity.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/cap/driver_like_*.json')):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'))
try:
    r=json.loads(open('gpurun_out/cap/capacity.json').read().strip().splitlines()[-1]); print('capacity', r['value'], r['ms_per_frame_at_value'], [(p['outputs'],p['ms_per_frame']) for p in r['probes']])
except Exception as e:
    print('capacity ERR', e); print(open('gpurun_out/cap/capacity.err').read()[-1500:])
PY
