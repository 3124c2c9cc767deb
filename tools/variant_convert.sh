#!/bin/bash
# A/B builds of the input converters:  tools/variant_convert.sh NAME [-DFLAG ...]   (here, no GPU needed; the twin of tools/variant.sh)
# compiles smr_convert.hip with the extra flags and links smelter_amd/variants/libsmr_hip.NAME.so from it and the other objects of the normal build
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m smelter_amd.build >/dev/null
mkdir -p smelter_amd/variants
obj=smelter_amd/variants/smr_convert.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function \
  -I include -I smelter_amd/csrc "$@" -x hip -c smelter_amd/csrc/smr_convert.hip -o $obj
others=$(ls smelter_amd/build/*.o | grep -v 'smr_convert.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o smelter_amd/variants/libsmr_hip.$name.so $obj $others
echo smelter_amd/variants/libsmr_hip.$name.so
