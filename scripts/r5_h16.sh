#!/bin/bash
# round 5: the 16-bit large-M kernel -- kernel tests in both storage flavours, the 16-bit model tests, then a same-box A/B of the f16 step
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r5m; mkdir -p $O
K='gemm_big_bf16_storage or (second_output and bf16)'
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/k_bf16.log 2>&1; echo "kernel tests bf16 rc=$? $(tail -1 $O/k_bf16.log)"
GAST_TEST_H16=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/k_f16.log 2>&1; echo "kernel tests f16 rc=$? $(tail -1 $O/k_f16.log)"
if [ -z "$SKIP_MODEL" ]; then
timeout 900 python -m pytest tests/test_f16_gpu.py -x -q -m gpu -p no:cacheprovider -k "not kernel_suite" > $O/m_f16.log 2>&1; echo "f16 model tests rc=$? $(tail -1 $O/m_f16.log)"
fi
bash scripts/ab_env.sh $O "old=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0" "new=GAST_HIP_DTYPE=f16"
