#!/bin/bash
# round 3: least tile rows per wave of k_ingest_wave (small launches: a nested node's single input) on configs[4] / configs[2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for m in 2 1; do
  for c in 4 2; do
    echo "== min rows $m config $c: $(SMR_INGEST_MIN_ROWS=$m python bench.py --config $c --no-target --no-cpu-baseline --latency-frames 100 --inflight 1 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernels'])")"
  done
done
