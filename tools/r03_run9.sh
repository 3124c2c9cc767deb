#!/bin/bash
# round 3: SMR_INGEST_MFMA_F16_NODE (exact converter + matrix-core resampler on the node texture) against the default on configs[2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for impl in auto mfma_node valu; do
  echo "== $impl: $(python bench.py --ingest $impl --no-target --no-cpu-baseline --latency-frames 200 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['config']['frames_per_s_one_in_flight'], j['kernels'], j['latency_ms']['p50'])")"
done
