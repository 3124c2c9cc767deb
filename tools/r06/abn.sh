#!/bin/bash
# alternating A/B of laboratory builds on one box:  tools/r06/abn.sh CONFIG ROUNDS NAME [NAME ...]   (3 s loops)
cd "$(dirname "$0")/../.."
c=$1; n=$2; shift 2
O=gpurun_out/abn; mkdir -p $O
for r in $(seq 1 $n); do
  for v in "$@"; do
    L=$PWD/smelter_amd/variants/libsmr_hip.$v.so
    SMR_LIB=$L timeout 600 python bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_${v}_c${c}_$r.json 2> $O/bench_${v}_c${c}_$r.err
    python - $O/bench_${v}_c${c}_$r.json $v <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], r['value'], (r.get('value_long') or {}).get('frames_per_s'), r['config'].get('frames_per_s_one_in_flight'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  done
done
