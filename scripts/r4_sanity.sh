#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm_big or wgrad or semch_agg" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-stock-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['parity']['pass'], d['variants']['f16']['ms_per_step'], d['variants']['f16']['parity']['pass'])"
