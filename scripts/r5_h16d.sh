#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r5o; mkdir -p $O
cd /tmp; GAST_HIP_DTYPE=f16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f16 -o f16 -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --no-parity --steps 10 --warmup 3 > $R/$O/prof_f16.log 2>&1
cd $R; cp $(find /tmp/prof_f16 -name '*kernel_trace.csv' | head -1) $O/f16_kernel_trace.csv 2>/dev/null
python scripts/trace_step.py $O/f16_kernel_trace.csv 3 > $O/f16_step_summary.txt 2>&1 || true
head -45 $O/f16_step_summary.txt | cut -c1-150
bash scripts/ab_env.sh $O "base="
