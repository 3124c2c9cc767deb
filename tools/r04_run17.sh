# round 4, run 17: float -> byte conversions that write the byte in place (v_cvt_i32_f32_sdwa) in the converter and the compositor's output conversion;
# the converter's RGBA8 / RGB12 stores as separate instantiations
mkdir -p gpurun_out/r04_17
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c2', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
for c in 3 4 1; do
timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
done
