#!/bin/bash
# upper bound of an LDS-DMA activation path in gemm_big: the profiling build with the activation loads / LDS writes switched off
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3q; mkdir -p $O
ABLATION=1 bash gast-net-3dposeestimation_amd/csrc/build.sh > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
for sh in g1s0 g1s1 g4s1 conv; do
  for ab in 0 16 24 1 25; do
    GAST_GEMM_BIG_ABLATE=$ab timeout 120 python scripts/gemm_big_ablate.py $sh 2>/dev/null | tail -1
  done
done | tee $O/ablate.txt
