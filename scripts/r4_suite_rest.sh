#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4_suite"; mkdir -p "$O"
for i in 1 2; do
timeout 2400 python -m pytest tests/test_reference_caller.py tests/test_f16_gpu.py -q -m gpu -p no:cacheprovider > "$O/gpu_suite_rest$i.log" 2>&1
echo "rest $i rc=$? : $(tail -1 $O/gpu_suite_rest$i.log)"
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "flat_gradient" 2>&1 | tail -1
done
grep -E "^E  |FAILED" $O/gpu_suite_rest*.log | head
