#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3f; mkdir -p $O
rm -f gpurun_out/model_parity_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|passed|failed|^E   " $O/tests.log | head -40
cp gpurun_out/model_parity_metrics.jsonl $O/metrics.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
