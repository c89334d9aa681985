#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
for i in 1 2 3 4; do
  GAST_TEST_H16=f16 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "test_gemm_bwd_second_output and bf16" 2>&1 | tail -1
done
for i in 1 2; do
  timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "test_gemm_bwd_second_output" 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "test_flat_gradient_buffer_accumulates_like_autograd" 2>&1 | tail -3
