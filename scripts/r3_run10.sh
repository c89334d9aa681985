#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3j; mkdir -p $O
rm -f gpurun_out/model_parity_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|passed|failed|^E   " $O/tests.log | head -30
cp gpurun_out/model_parity_metrics.jsonl $O/metrics.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['parity']['pass'], d['roofline']['frac'], d['roofline']['traffic_source'][:40], d.get('module_graph',{}).get('ms_per_step'))"
