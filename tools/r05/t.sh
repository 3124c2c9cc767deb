cd ${GRAFT_REPO_ROOT:-$(pwd)}; timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "assorted" 2>&1 | tail -15
