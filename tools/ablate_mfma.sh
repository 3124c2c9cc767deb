#!/bin/bash
# GPU box: times the (3,2) build of k_ingest_mfma with phases compiled out (library built with SMR_ABLATION_BUILDS=1).
# usage: tools/ablate_mfma.sh "<ablate values>" [iw ih dw dh n]
A=${1:-"0 1 2 4 5 6 7 16 23 64 128"}; shift
G=${@:-1920 1080 1280 720 8}
for a in $A; do
  echo -n "ablate $a: "
  SMR_ABLATE=$a python tools/check_mfma.py $G 2>&1 | grep "kcycles\|launch" | tail -2 | tr '\n' ' '
  echo
done
