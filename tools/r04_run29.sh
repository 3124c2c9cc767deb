# round 4, run 29: three / four frames in flight with smaller kernels (fewer resampler pieces), last look
for v in "if2:--inflight 2" "if3:--inflight 3" "if4:--inflight 4"; do
  n=${v%%:*}; a=${v#*:}
  for cap in 0 4 3; do
  SMR_INGEST_WG_PER_CU=$cap timeout 200 python bench.py $a --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 50 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$n cap $cap', r['value'], 'fps')"
  done
done
