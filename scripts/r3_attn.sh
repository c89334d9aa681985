#!/bin/bash
# round 3: compile-time-J attention wave kernels -- unroll factors (compile time) and JT on/off (run time) on one box
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3p; mkdir -p $O
AB_TESTS="attention or attn" bash scripts/ab_variants.sh $O "u4:" "u17:-DGAST_ATTN_PROD_UNROLL=17" "u2:-DGAST_ATTN_PROD_UNROLL=2" "u8:-DGAST_ATTN_PROD_UNROLL=8" > $O/ab.txt 2>&1
EXTRA_FLAGS="" bash gast-net-3dposeestimation_amd/csrc/build.sh > $O/build_final.log 2>&1
bash scripts/ab_env.sh $O "jt0:GAST_ATTN_JT=0" "jt1:GAST_ATTN_JT=1" >> $O/ab.txt 2>&1
cat $O/ab.txt; tail -3 $O/tests_u4.log
