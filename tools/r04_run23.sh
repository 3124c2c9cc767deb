# round 4, run 23: converter with 24-bit row offsets (no quarter-rate 32 x 32 multiplies) and one instantiation per node format (each packs its own bytes only)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for i in 1 2; do
for c in 2 3 1; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
