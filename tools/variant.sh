#!/bin/bash
# Laboratory builds of the library (here, no GPU needed):  tools/variant.sh NAME [-DFLAG ...]
# Compiles every HIP source with -DSMR_LAB (the A/B knobs read from the environment at context creation, the fused-conversion builds of
# k_ingest_wave, ablation / timing hooks) plus the extra flags — e.g. -DCV_TIMING, -DCV_ABL=7, -DSMR_WAVE_TIMING=1 — and links
# smelter_amd/variants/libsmr_hip.NAME.so with the host objects of the normal build.  On the GPU box SMR_LIB=<that file> picks it
# (tools/ab.sh NAME... benches several).  A product build (python -m smelter_amd.build) has none of this.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m smelter_amd.build >/dev/null
dir=smelter_amd/variants/$name.d
mkdir -p $dir
pids=()
for src in smelter_amd/csrc/*.hip; do
  obj=$dir/$(basename $src).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function \
    -I include -I smelter_amd/csrc -DSMR_LAB "$@" -x hip -c $src -o $obj &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
host=$(ls smelter_amd/build/*.cpp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o smelter_amd/variants/libsmr_hip.$name.so $dir/*.o $host -ldl
echo smelter_amd/variants/libsmr_hip.$name.so
