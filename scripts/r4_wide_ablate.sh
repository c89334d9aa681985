#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
export GAST_HIP_DTYPE=bf16x3 GAST_WGRAD_X3_TILE=256
echo "== wide"; timeout 300 python scripts/wgrad_multi_bench.py s1 s2 2>&1 | tail -2
for n in 1 2 3 4; do
  echo "== wide WG_ABLATE=$n"; GAST_HIP_LIB_EXPERIMENT=wab$n timeout 300 python scripts/wgrad_multi_bench.py s1 s2 2>&1 | tail -2
done
