#!/bin/bash
# round 3: where configs[4]'s compositor time goes — SMR_ABLATE bits << 8: 1 compositing tiles stop after the list copy, 2 after the
# touch record, 8 base layer only, 16 no compositing (TC_FULL) tiles, 32 no sampled tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for a in ${ABLATES:-0 4096 8192 12288}; do
  echo "== ablate $a: $(SMR_ABLATE=$a python bench.py --config 4 --no-target --no-cpu-baseline --latency-frames 100 --inflight 1 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernels'])")"
done
