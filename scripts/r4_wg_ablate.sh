#!/bin/bash
# wgrad_x3_pipe ablation builds (WG_ABLATE=1..4: no loads / cache-resident loads / no MFMA / no conversion + LDS write) on the three stage job sets
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
export GAST_HIP_DTYPE=bf16x3
echo "== baseline"; timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
for n in 1 2 3 4; do
  echo "== WG_ABLATE=$n"; GAST_HIP_LIB_EXPERIMENT=wg$n timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
done
