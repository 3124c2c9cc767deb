# round 4, run 13: how the pipelined frame rate responds to less intermediate traffic (direct output) and to CUs left free by the resampler
# (SMR_INGEST_RESERVE_CUS shrinks its grid; the converter / compositor of the other frame in flight take the room)
mkdir -p gpurun_out/r04_13
run() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 150 python bench.py --no-cpu-baseline --no-target --no-long --steps 300 --warmup 30 --latency-frames 200 "$@" 2>/dev/null > gpurun_out/r04_13/$tag.json
  python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
try:
    r=json.load(open(f'gpurun_out/r04_13/{t}.json'))
    print(f"{t:22s} {r['value']:9.1f} frames/s  {r['config'].get('frames_per_s_one_in_flight',0):9.1f} (1 in flight)  p50 {r['latency_ms']['p50']:.4f} ms  kernels us:", ', '.join(f"{k} {v['avg_us']}" for k,v in r['kernels'].items()))
except Exception as e:
    print(t, 'failed', e)
PY
}
run base X=1 --
run direct X=1 -- --direct-output
run base_b X=1 --
for r in 16 32 64 96; do
  run reserve$r SMR_INGEST_RESERVE_CUS=$r --
  run direct_reserve$r SMR_INGEST_RESERVE_CUS=$r -- --direct-output
done
run direct_inflight3 X=1 -- --direct-output --inflight 3
run c3_base X=1 -- --config 3
run c3_direct X=1 -- --config 3 --direct-output
run c3_reserve64 SMR_INGEST_RESERVE_CUS=64 -- --config 3
