#!/bin/bash
# full GPU validation: the suite, smoke, default bench (logs under gpurun_out/$1)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/${1:-r3full}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.log; tail -3 $O/smoke.log; python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['parity']['pass'], d['parity']['vs_fp32_hip'], d['parity'].get('vs_cpu_reference_restatement'))
P
