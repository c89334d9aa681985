timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2
echo "== MW=2"; timeout 300 python scripts/gemm_table.py bf16x3 2>&1 | grep -E "gemm" | grep -v "M=2176"
echo "== MW=4"; GAST_GEMM_BIG_MW=4 timeout 300 python scripts/gemm_table.py bf16x3 2>&1 | grep -E "gemm" | grep -v "M=2176"
for e in "X=1" "GAST_GEMM_BIG_MW=4" "X=1" "GAST_GEMM_BIG_MW=4 GAST_GEMM_BIG_MW_MIN_K=700"; do echo "== $e"; env $e timeout 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-parity 2>&1 | tail -1 | cut -c90-200; done
