#!/bin/bash
# usage: r4_wide_var.sh name1 name2 ... : time libgast_hip_<name>.so variants ("base" = the production library) of the wide weight-gradient kernel on stages s1, s2
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
export GAST_HIP_DTYPE=bf16x3
for n in "$@"; do
  echo "== $n"
  if [ "$n" = base ]; then timeout 300 python scripts/wgrad_multi_bench.py s1 s2 2>&1 | tail -2
  else GAST_HIP_LIB_EXPERIMENT=$n timeout 300 python scripts/wgrad_multi_bench.py s1 s2 2>&1 | tail -2; fi
done
