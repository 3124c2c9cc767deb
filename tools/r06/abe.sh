#!/bin/bash
# alternating A/B with an environment per leg:  tools/r06/abe.sh CONFIG ROUNDS "NAME[:VAR=VAL[,VAR=VAL]]" ...
cd "$(dirname "$0")/../.."
c=$1; n=$2; shift 2
O=gpurun_out/abe; mkdir -p $O
for r in $(seq 1 $n); do
  for leg in "$@"; do
    v=${leg%%:*}; e=""; [ "$leg" != "$v" ] && e=$(echo "${leg#*:}" | tr ',' ' ')
    L=$PWD/smelter_amd/variants/libsmr_hip.$v.so
    tag=$(echo "$leg" | tr ':=,' '___')
    env SMR_LIB=$L $e timeout 600 python bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_${tag}_c${c}_$r.json 2> $O/bench_${tag}_c${c}_$r.err
    python - $O/bench_${tag}_c${c}_$r.json "$leg" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], r['value'], (r.get('value_long') or {}).get('frames_per_s'), r['config'].get('frames_per_s_one_in_flight'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  done
done
