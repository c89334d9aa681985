#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3c; mkdir -p $O
AB_TESTS="gemm or wgrad or bn or norm" bash scripts/ab_variants.sh $O "base:" "prio:-DGAST_MFMA_PRIO" "wprio:-DGAST_WGRAD_PRIO" "both:-DGAST_MFMA_PRIO -DGAST_WGRAD_PRIO"
# leave the library in its base state and run the model-level tests that touch the changed host paths
EXTRA_FLAGS="" bash gast-net-3dposeestimation_amd/csrc/build.sh > $O/build_final.log 2>&1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_modules_gpu.py tests/test_inference_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "(golden and not bf16-) or trajectory or module or shape243 or midsize" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|passed|failed" $O/tests.log | head -20
