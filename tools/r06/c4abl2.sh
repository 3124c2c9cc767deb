#!/bin/bash
# which step of the list compositor a band's time goes to: configs[4] frozen in motion, sampled tiles off (SMR_ABLATE 8192), + 2048 = start values only (no list,
# no compositing), + 1024 = the tile's own start layer only (no claims by the layers above it), + 512 = list load only
cd "$(dirname "$0")/../.."
L=$PWD/smelter_amd/variants/libsmr_hip.$1.so
for ab in 8192 10240 9216 11264 8704 12288; do SMR_LIB=$L SMR_ABLATE=$ab timeout 300 python tools/r06/c4probe.py 200 0.5 2>&1 | tail -1; done
