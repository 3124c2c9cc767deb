mkdir -p gpurun_out
for a in 0 256 512 4096 8192; do
  SMR_ABLATE=$a timeout 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --latency-frames 5 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ablate $a', r['value'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
