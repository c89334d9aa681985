#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
for v in "GAST_AGG_BWD_LDS=1 GAST_AGG_BWD_BLOCKS=512" "GAST_HIP_LIB_EXPERIMENT=u4 GAST_AGG_BWD_LDS=1 GAST_AGG_BWD_BLOCKS=512" "GAST_HIP_LIB_EXPERIMENT=u4 GAST_AGG_BWD_LDS=1 GAST_AGG_BWD_BLOCKS=768" "GAST_HIP_LIB_EXPERIMENT=u3 GAST_AGG_BWD_LDS=1 GAST_AGG_BWD_BLOCKS=512"; do
  rm -rf /tmp/prof
  env $v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 6 --warmup 2 > /tmp/tr.log 2>&1
  T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python scripts/trace_step.py $T 3 | grep -E "steps=|semch_agg_bwd"
  python scripts/trace_timeline.py $T /tmp/tl.txt > /dev/null 2>&1; grep semch_agg_bwd /tmp/tl.txt | head -3
done
