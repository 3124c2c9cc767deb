#!/bin/bash
# round 6: (1) v_cvt_pk_u8_f32 under round-toward-zero; (2) where direct output loses its time (laboratory builds without its arithmetic / its stores);
# (3) two copy tiles per compositor workgroup now that the conversion is cheap; (4) kernel trace of the product build, one frame in flight
cd "$(dirname "$0")/../.."
O=gpurun_out/ab3; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt_mode tools/ubench/cvt_pk_u8_mode.hip 2>/dev/null && /tmp/cvt_mode > $O/cvt_mode.txt 2>&1
cat $O/cvt_mode.txt
for n in d0 d1 d2 d3; do
  SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.$n.so timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 --direct-output > $O/bench_${n}_if1.json 2> $O/bench_${n}_if1.err
done
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.ct2.so timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_ct2.json 2> $O/bench_ct2.err
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.ct2.so timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --no-long --inflight 1 > $O/bench_ct2_if1.json 2> $O/bench_ct2_if1.err
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-target --long-seconds 3 > $O/bench_product.json 2> $O/bench_product.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace -o t -- python $OLDPWD/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-target --no-long --inflight 1 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python - <<'PY'
import json,glob,csv,os
for f in sorted(glob.glob('gpurun_out/ab3/bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], r['value'], (r.get('value_long') or {}).get('frames_per_s'), {k:v['avg_us'] for k,v in (r.get('kernels') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
for root,_,files in os.walk('gpurun_out/ab3/trace'):
    for f in files:
        if f.endswith('kernel_stats.csv'):
            for row in list(csv.DictReader(open(os.path.join(root,f))))[:6]:
                print(row.get('Name','')[:40], row.get('Calls'), row.get('AverageNs'))
PY
