#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r6_suite; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -15 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
