#!/bin/bash
# launch-shape sweep under two lanes with round 6's kernels (laboratory build: host knobs from the environment)
cd "$(dirname "$0")/../.."
O=gpurun_out/sweep; mkdir -p $O
L=$PWD/smelter_amd/variants/libsmr_hip.lab0.so
run() { n=$1; shift; env SMR_LIB=$L "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_$n.json 2> $O/bench_$n.err; }
run default_a
run conv4 SMR_CONVERT_WG_PER_CU=4
run conv5 SMR_CONVERT_WG_PER_CU=5
run conv3 SMR_CONVERT_WG_PER_CU=3
run res5 SMR_INGEST_WG_PER_CU=5
run res6 SMR_INGEST_WG_PER_CU=6
run res3 SMR_INGEST_WG_PER_CU=3
run res6conv4 SMR_INGEST_WG_PER_CU=6 SMR_CONVERT_WG_PER_CU=4
run default_b
n=if3; SMR_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 --inflight 3 > $O/bench_$n.json 2> $O/bench_$n.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sweep/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
