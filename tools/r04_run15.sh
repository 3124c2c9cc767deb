# round 4, run 15: more, shorter pieces in the resampler (SMR_INGEST_OVERSUB), reserves with three frames in flight, and what the compositor's
# grid looks like with direct output + the compacted list
mkdir -p gpurun_out/r04_15
SMR_DEBUG_INGEST=1 timeout 120 python bench.py --direct-output --steps 8 --warmup 4 --no-cpu-baseline --no-target --no-long --latency-frames 5 2>&1 | grep -E "k_compose_output|tile classes" | tail -14
run() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 150 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 "$@" 2>/dev/null > gpurun_out/r04_15/$tag.json
  python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
try:
    r=json.load(open(f'gpurun_out/r04_15/{t}.json'))
    print(f"{t:22s} {r['value']:9.1f} frames/s  {r['config'].get('frames_per_s_one_in_flight',0):9.1f} (1 in flight)  p50 {r['latency_ms']['p50']:.4f} ms  kernels us:", ', '.join(f"{k} {v['avg_us']}" for k,v in r['kernels'].items()))
except Exception as e:
    print(t, 'failed', e)
PY
}
run base X=1 --
run oversub1.5 SMR_INGEST_OVERSUB=1.5 --
run oversub2 SMR_INGEST_OVERSUB=2 --
run oversub3 SMR_INGEST_OVERSUB=3 --
run oversub4 SMR_INGEST_OVERSUB=4 --
run reserve64 SMR_INGEST_RESERVE_CUS=64 --
run reserve96 SMR_INGEST_RESERVE_CUS=96 --
run inflight3_reserve64 SMR_INGEST_RESERVE_CUS=64 -- --inflight 3
run inflight3_reserve96 SMR_INGEST_RESERVE_CUS=96 -- --inflight 3
run inflight3_oversub2 SMR_INGEST_OVERSUB=2 -- --inflight 3
run base_b X=1 --
run c3_base X=1 -- --config 3
run c3_oversub2 SMR_INGEST_OVERSUB=2 -- --config 3
run c3_oversub3 SMR_INGEST_OVERSUB=3 -- --config 3
run c3_reserve64 SMR_INGEST_RESERVE_CUS=64 -- --config 3
run c3_inflight3_reserve64 SMR_INGEST_RESERVE_CUS=64 -- --config 3 --inflight 3
