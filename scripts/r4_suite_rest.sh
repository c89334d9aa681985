#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4_suite"; mkdir -p "$O"
timeout 2400 python -m pytest tests/test_reference_caller.py tests/test_streaming_gpu.py -q -m gpu > "$O/gpu_suite_rest.log" 2>&1
echo "rest rc=$? : $(tail -1 $O/gpu_suite_rest.log)"
grep -E "^E |FAILED|Error" "$O/gpu_suite_rest.log" | head -20
