#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r05_final
bash tools/gpu_tests.sh r05_final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_final/bench_driver_cmd.json 2> gpurun_out/r05_final/bench_driver_cmd.err
echo "bench wall seconds: $(( $(date +%s) - S ))"
python - <<PY
import json
r=json.loads(open("gpurun_out/r05_final/bench_driver_cmd.json").read().strip().splitlines()[-1])
print(r["metric"], r["value"], r["value_long"]["frames_per_s"], r["ms_per_step"], "stale:", r["roofline"]["traffic_stale"], "traffic x", r["roofline"]["traffic_over_algorithmic"], "issue", r["roofline"]["issue"].get("frame_over_floor"))
print("target", r["target"]["frames_per_s"], "sharded", r["sharded_one_rank"].get("frames_per_s"), "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["spread"])
PY
