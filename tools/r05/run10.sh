#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_run10
mkdir -p $O
cd $R
bash tools/gpu_tests.sh r05_product
SMR_LIB=$R/smelter_amd/variants/libsmr_hip.lab.so timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernel_selection.py -m gpu -q > $O/pytest_lab.log 2>&1; echo "lab pytest rc $?"; tail -3 $O/pytest_lab.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
