# round 4, run 14: (a) compositor over the classifier's compacted tile list: tests + direct output again; (b) frames in flight on disjoint
# sets of compute units (SMR_LANE_CU_SPLIT: a CU-masked stream per lane) and larger reserves of the resampler's grid
mkdir -p gpurun_out/r04_14
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
run() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 150 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 "$@" 2>/dev/null > gpurun_out/r04_14/$tag.json
  python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
try:
    r=json.load(open(f'gpurun_out/r04_14/{t}.json'))
    print(f"{t:22s} {r['value']:9.1f} frames/s  {r['config'].get('frames_per_s_one_in_flight',0):9.1f} (1 in flight)  p50 {r['latency_ms']['p50']:.4f} ms  kernels us:", ', '.join(f"{k} {v['avg_us']}" for k,v in r['kernels'].items()))
except Exception as e:
    print(t, 'failed', e)
PY
}
run base X=1 --
run direct_compact X=1 -- --direct-output
run direct_nocompact SMR_COMPOSE_COMPACT=0 -- --direct-output
run reserve128 SMR_INGEST_RESERVE_CUS=128 --
run reserve160 SMR_INGEST_RESERVE_CUS=160 --
run split2_halves SMR_LANE_CU_SPLIT=2:halves --
run split2_interleave SMR_LANE_CU_SPLIT=2:interleave --
run split2_xcd SMR_LANE_CU_SPLIT=2:xcd --
run split3_interleave SMR_LANE_CU_SPLIT=3:interleave -- --inflight 3
run split4_xcd SMR_LANE_CU_SPLIT=4:xcd -- --inflight 4
run base_b X=1 --
run c3_base X=1 -- --config 3
run c3_reserve96 SMR_INGEST_RESERVE_CUS=96 -- --config 3
run c3_reserve128 SMR_INGEST_RESERVE_CUS=128 -- --config 3
run c3_split2_interleave SMR_LANE_CU_SPLIT=2:interleave -- --config 3
run c3_split2_xcd SMR_LANE_CU_SPLIT=2:xcd -- --config 3
run c4_base X=1 -- --config 4
run c4_split2_xcd SMR_LANE_CU_SPLIT=2:xcd -- --config 4
