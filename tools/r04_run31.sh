# round 4, run 31: the converter's LDS tables from a per-context device table instead of a division per luma byte per workgroup
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|ERROR|rror" | tail -4
for i in 1 2; do
for c in 2 3; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
