#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r5a; mkdir -p $O
bash scripts/stress_kernel_suite.sh $O/stress 3 2>&1 | tee $O/stress.log
timeout 900 python -m pytest tests/test_deterministic_gpu.py -m gpu -q -p no:cacheprovider -x > $O/det.log 2>&1; echo "det rc=$?"; tail -15 $O/det.log
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench', d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))"
