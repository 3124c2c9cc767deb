# round 4, run 21: TC_SELECT (seam tiles as per-pixel copies from the topmost opaque 1:1 layer): tests, then configs[4] / [1] / [2] with and without
mkdir -p gpurun_out/r04_21
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py tests/test_gpu_reference_scenes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6
for s in 1 0; do
for c in 4 1 2; do
SMR_COMPOSE_SELECT=$s timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('select $s c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
