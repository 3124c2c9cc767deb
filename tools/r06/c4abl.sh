#!/bin/bash
# configs[4] frozen in motion, one laboratory build, the compositor's tile classes switched off one by one (SMR_ABLATE bits 8..: 32 sampled, 16 composited, 64 copy)
cd "$(dirname "$0")/../.."
L=$PWD/smelter_amd/variants/libsmr_hip.$1.so
for ab in 0 8192 4096 16384 12288 28672; do SMR_LIB=$L SMR_ABLATE=$ab timeout 300 python tools/r06/c4probe.py 200 0.5 2>&1 | tail -1; done
