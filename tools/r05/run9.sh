#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run9
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_renderer.py tests/test_gpu_fused.py tests/test_gpu_comm.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 1500 $O/bench_default.err
python - <<PY
import json
r=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "long", r["value_long"], "one", r["config"]["frames_per_s_one_in_flight"])
print("roofline", json.dumps({k:v for k,v in r["roofline"].items() if k not in ("limiter",)})[:1500])
print("cpu", json.dumps(r.get("cpu_baseline"))[:600])
print("target", json.dumps(r.get("target"))[:400])
print("sharded", json.dumps(r.get("sharded_one_rank"))[:900])
PY
