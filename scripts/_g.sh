timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm and not optin" 2>&1 | tail -3
echo "== split big"; timeout 300 python scripts/gemm_table.py bf16x3 2>&1 | grep -E "M=2176|total" | grep -v wgrad
echo "== old"; GAST_GEMM_BIG_SPLIT=0 timeout 300 python scripts/gemm_table.py bf16x3 2>&1 | grep -E "M=2176|total" | grep -v wgrad
