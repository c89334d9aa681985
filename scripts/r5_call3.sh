#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r5c; mkdir -p $O
timeout 240 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "lazy or fused" > $O/lazy_tests.log 2>&1; rc=$?
echo "lazy kernel tests rc=$rc"; tail -25 $O/lazy_tests.log
if [ $rc -ne 0 ]; then exit 0; fi
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "not optin" > $O/kernels.log 2>&1; echo "kernel suite rc=$? $(tail -1 $O/kernels.log)"
for lz in 1 0; do
  GAST_LAZY_BN=$lz timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > $O/bench_lazy$lz.json 2> $O/bench_lazy$lz.err
  python -c "import json;d=json.loads(open('$O/bench_lazy$lz.json').read().strip().splitlines()[-1]);print('lazy=$lz', d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d['parity'].get('pass'), d.get('forward_only',{}).get('ms'), d.get('kernels_per_step'))" || tail -5 $O/bench_lazy$lz.err
done
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x > $O/model.log 2>&1; echo "model suite rc=$? $(tail -1 $O/model.log)"
