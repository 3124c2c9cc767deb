# round 4, run 28: fewer, longer pieces in k_ingest_wave (a piece re-computes the chunks above its first tile row): SMR_INGEST_WG_PER_CU caps the resident workgroups the
# piece count is sized for
for cap in 0 5 4 3; do
  for c in 2 3 1; do
  SMR_INGEST_WG_PER_CU=$cap SMR_DEBUG_INGEST=1 timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 8 --warmup 4 --latency-frames 4 2>&1 | grep "k_ingest_wave\[" | sort | uniq -c | sort -rn | head -1 | cut -c1-200
  SMR_INGEST_WG_PER_CU=$cap timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('cap $cap c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
  done
done
