for r in 0 8 16 24 32; do
  SMR_INGEST_RESERVE_CUS=$r timeout 120 python bench.py --no-cpu-baseline --steps 300 --warmup 30 --latency-frames 5 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('reserve $r', r['value'], {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
