#!/bin/bash
# Ablation matrix of gemm_big on the step's shapes (profiling library libgast_hip_abl.so: -DGAST_GEMM_BIG_ABLATION).  bits: 1 no MFMA,
# 2 no fragment reads (and no MFMA), 4 no weight DMA, 8 no activation loads, 16 no activation LDS writes (no prologue / split), 32 no epilogue
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
export GAST_HIP_LIB_EXPERIMENT=abl
for sh in g4s1 g1s1 g1s0 conv; do
  for a in 0 32 1 3 4 8 16 24 28 29 31 63; do
    GAST_GEMM_BIG_ABLATE=$a python scripts/gemm_big_ablate.py $sh 2>&1 | tail -1
  done
done
