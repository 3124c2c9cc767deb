for c in 1 2 3; do
  for n in 1 3; do
  timeout 280 python bench.py --no-cpu-baseline --config $c --inflight $n --steps 200 --warmup 20 --latency-frames 50 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('config $c inflight $n', r['value'], {k:v['avg_us'] for k,v in r['kernels'].items()}, r['latency_ms']['p50'], r['frame']['frac_of_hbm_peak'])"
  done
done
