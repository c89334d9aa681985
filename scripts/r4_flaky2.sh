#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
for i in 1 2 3; do
  GAST_TEST_H16=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "(bf16 or dt1 or dtype1 or float16 or out_f32) and not optin and not fp8 and not x3" 2>&1 | tail -3
done
GAST_TEST_H16=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "(bf16 or dt1 or dtype1 or float16 or out_f32) and not optin and not fp8 and not x3" -v 2>&1 | grep -E "PASSED|FAILED" | head -30
