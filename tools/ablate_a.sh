for a in 0 1 2 4 8 128 3 7 135; do
  SMR_ABLATE=$a timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --latency-frames 5 --inflight 1 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ablate $a', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
