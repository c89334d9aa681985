#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3e; mkdir -p $O
bash scripts/ab_env.sh $O "base:GAST_X=0" "aggf2:GAST_AGG_FWD_JSPLIT=2" "aggf4:GAST_AGG_FWD_JSPLIT=4" "aggb2:GAST_AGG_BWD_JSPLIT=2" "aggb4:GAST_AGG_BWD_JSPLIT=4" "aggfb4:GAST_AGG_FWD_JSPLIT=4 GAST_AGG_BWD_JSPLIT=4" "base2:GAST_X=0"
for v in "GAST_AGG_FWD_JSPLIT=4 GAST_AGG_BWD_JSPLIT=4" "GAST_AGG_FWD_JSPLIT=2 GAST_AGG_BWD_JSPLIT=2"; do
  env $v timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "semch or agg" 2>&1 | tail -2
done
