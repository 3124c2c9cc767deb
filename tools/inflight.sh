for n in 1 2 3 4; do
  timeout 120 python bench.py --no-cpu-baseline --steps 300 --warmup 30 --latency-frames 5 --inflight $n 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('inflight $n', r['value'], r['ms_per_step'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
