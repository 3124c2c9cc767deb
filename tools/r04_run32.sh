# round 4, run 32: the compositor as two launches (SMR_COMPOSE_SPLIT=1): band list + sampled tiles first (106 VGPRs), then the plain copies in a kernel of their
# own (45 VGPRs, no LDS: eight waves per SIMD instead of four)
SMR_COMPOSE_SPLIT=1 timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py tests/test_gpu_reference_scenes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|ERROR|rror" | tail -4
for s in 0 1 0 1; do
for c in 2 4 3; do
SMR_COMPOSE_SPLIT=$s timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 300 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('split $s c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'p50', r['latency_ms']['p50'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
