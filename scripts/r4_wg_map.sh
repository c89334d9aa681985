#!/bin/bash
# wgrad_x3_pipe lane -> (row block, column quad) maps: WG_MAP=0 (row block fastest), 1 (a quad of lanes = 64 B of a row), 2 (8 lanes = 128 B)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
export GAST_HIP_DTYPE=bf16x3
echo "== baseline"; timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
for n in 1 2 3; do
  echo "== WG_MAP=$n"; GAST_HIP_LIB_EXPERIMENT=map$n timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
done
echo "== baseline"; timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
