#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r5n; mkdir -p $O
K='wide_wgrad_bf16 or (test_wgrad and bf16) or gemm_big_bf16_storage'
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/k_bf16.log 2>&1; echo "kernel tests bf16 rc=$? $(tail -1 $O/k_bf16.log)"
GAST_TEST_H16=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/k_f16.log 2>&1; echo "kernel tests f16 rc=$? $(tail -1 $O/k_f16.log)"
timeout 900 python -m pytest tests/test_f16_gpu.py -x -q -m gpu -p no:cacheprovider -k "not kernel_suite" > $O/m_f16.log 2>&1; echo "f16 model tests rc=$? $(tail -1 $O/m_f16.log)"
bash scripts/ab_env.sh $O "old=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0,GAST_WGRAD_H16_WIDE=0" "wg=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0" "both=GAST_HIP_DTYPE=f16"
