#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r5d; mkdir -p $O
timeout 240 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "lazy or fused" > $O/lazy_tests.log 2>&1; rc=$?
echo "lazy kernel tests rc=$rc $(tail -1 $O/lazy_tests.log)"
if [ $rc -ne 0 ]; then tail -30 $O/lazy_tests.log; exit 0; fi
b() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d['parity'].get('pass'), d.get('forward_only',{}).get('ms'))" || tail -5 $O/bench_$name.err
}
b all_on GAST_LAZY_BN=1
b lazy_off GAST_LAZY_BN=0
b lazy_off_x0_off GAST_LAZY_BN=0 GAST_LAZY_X0=0
b on_no_aggfuse GAST_LAZY_BN=1 GAST_FUSE_AGG_BN=0 GAST_FUSE_EXPAND_BN=0
b all_on2 GAST_LAZY_BN=1
