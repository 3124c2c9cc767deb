# round 4, run 26: one-gather encode (6.5 KB of buckets) against the two-gather encode (2.7 KB) now that the node-texture builds allocate no staging area:
# LDS per workgroup and resident workgroups per CU of every build (SMR_DEBUG_INGEST), frames/s and stage times per config
for name in base enc0; do
  lib=smelter_amd/libsmr_hip.so
  [ "$name" != base ] && lib=smelter_amd/variants/libsmr_hip.$name.so
  for c in 2 3 1 4; do
  SMR_DEBUG_INGEST=1 SMR_LIB=$PWD/$lib timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 8 --warmup 4 --latency-frames 4 2>&1 | grep "k_ingest_wave\[" | sort | uniq -c | sort -rn | head -2 | cut -c1-200
  for i in 1 2; do
  SMR_LIB=$PWD/$lib timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$name c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
  done
  done
done
