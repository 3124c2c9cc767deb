# round 4, run 24: compositor with 24-bit row offsets (copy, select and sampled paths, output conversion)
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py tests/test_gpu_reference_scenes.py tests/test_gpu_layout.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6
for i in 1 2; do
for c in 2 4 1; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
