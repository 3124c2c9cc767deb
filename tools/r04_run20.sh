# round 4, run 20: the converter with every load of a block in flight before its first store, under the three workgroup orders
mkdir -p gpurun_out/r04_20
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for o in 0 1 2 0 2; do
SMR_CONVERT_ORDER=$o timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('order $o c2', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
for o in 0 1 2; do
for c in 3 4 1; do
SMR_CONVERT_ORDER=$o timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('order $o c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
done
