#!/bin/bash
# round 6: what the pipelined frame is sensitive to — the same bench on the baseline library, the current one and ablation builds (laboratory builds:
# converter without stores / without per-pixel arithmetic, resampler without stores / without table gathers / without pass 2 + encode)
cd "$(dirname "$0")/../.."
O=gpurun_out/ab2; mkdir -p $O
for n in product base cvabl1 cvabl8 wabl16 wabl1 wabl8; do
  lib=smelter_amd/variants/libsmr_hip.$n.so; [ $n = product ] && lib=smelter_amd/libsmr_hip.so
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_$n.json 2> $O/bench_$n.err
  SMR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 > $O/bench_${n}_if1.json 2> $O/bench_${n}_if1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab2/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
