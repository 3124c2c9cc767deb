# round 4, run 16: two copy tiles per compositor workgroup (texel loads of both in flight), variant build copy2
mkdir -p gpurun_out/r04_16
SMR_LIB=$PWD/smelter_amd/variants/libsmr_hip.copy2.so timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_renderer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
for i in 1 2; do bash tools/ab.sh copy2; done
bash tools/ab.sh --config 4 copy2
bash tools/ab.sh --config 3 copy2
