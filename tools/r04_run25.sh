# round 4, run 25: k_ingest_wave's encode with one table gather per value (bucket entry = code | offset of the threshold inside the bucket)
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|ERROR|rror" | tail -5
for i in 1 2; do
for c in 2 3 1 4; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
