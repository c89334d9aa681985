#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r5m; mkdir -p $O
GAST_TEST_H16=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "second_output and bf16" > $O/k_f16b.log 2>&1; echo "kernel tests f16 rc=$? $(tail -1 $O/k_f16b.log)"
bash scripts/ab_env.sh $O "old=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0" "new=GAST_HIP_DTYPE=f16"
cd /tmp; GAST_HIP_DTYPE=f16 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f16 -o f16 -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --no-parity --steps 10 --warmup 3 > $R/$O/prof_f16.log 2>&1
cd $R; ls $O/prof_f16 | head; f=$(ls $O/prof_f16/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-160
