#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4_suite"; mkdir -p "$O"
timeout 2400 python -m pytest tests/ -q -m gpu > "$O/gpu_suite.log" 2>&1
echo "gpu suite rc=$? : $(tail -1 $O/gpu_suite.log)"
grep -E "^E |FAILED|Error" "$O/gpu_suite.log" | head -20
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
