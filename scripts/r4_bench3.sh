#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['parity']['pass'], d['forward_only']['ms'])"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
