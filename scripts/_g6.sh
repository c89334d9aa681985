timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or x3_image" 2>&1 | tail -2
for sh in g4s1 g1s0 conv g1s1 k2048; do python scripts/gemm_big_ablate.py $sh 2>&1 | grep ablate; done
timeout 300 python scripts/gemm_table.py bf16x3 2>&1 | grep -E "^gemm|^wgrad|^total|fault"
