#!/bin/bash
# round 3: fp16-pair forward GEMMs (GAST_F32X3H) -- kernel tests, the bf16x3 model tests, A/B against bf16 pairs on one box
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3n; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or x3_image" 2>&1 | tail -15 > $O/kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_inference_gpu.py -q -k "bf16x3" 2>&1 | tail -25 > $O/model.log
bash scripts/ab_env.sh $O "f16:GAST_X3_FWD=f16" "bf16:GAST_X3_FWD=bf16" "f16b:GAST_X3_FWD=f16" > $O/ab.txt 2>&1
cat $O/kernels.log $O/model.log $O/ab.txt
