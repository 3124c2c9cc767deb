# round 4, run 27: node textures at scales around 2 (configs[1], configs[4]) on <8, 2> class builds of k_ingest_wave instead of the generic build; RGB12 nodes for them
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|ERROR|rror" | tail -5
for v in "node82=0:SMR_WAVE_NODE82=0" "node82=1:SMR_WAVE_NODE82=1" "node82+rgb12:SMR_RGB12_CLS82=1"; do
  n=${v%%:*}; e=${v#*:}
  for c in 1 4; do
  env $e SMR_DEBUG_INGEST=1 timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 8 --warmup 4 --latency-frames 4 2>&1 | grep "k_ingest_wave\[" | sort | uniq -c | sort -rn | head -2 | cut -c1-200
  for i in 1 2; do
  env $e timeout 200 python bench.py --config $c --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$n c$c', r['value'], 'fps', r['config']['frames_per_s_one_in_flight'], 'serial', {k:v['avg_us'] for k,v in r['kernels'].items()})"
  done
  done
done
SMR_RGB12_CLS82=1 timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|ERROR|rror" | tail -3
