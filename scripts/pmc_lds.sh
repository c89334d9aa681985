#!/bin/bash
# LDS bank-conflict counters of every kernel of the eager training step (rocprofv3 --pmc, its own run): OUT/pmc_lds_<kernel>.txt
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; mkdir -p "$O"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_lds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph --no-twin --no-f16 --no-stock-baseline > $O/pmc_lds.log 2>&1
F=$(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1)
for k in "gemm_big_kernel<0" "gemm_big_kernel<1" "gemm_big_kernel<2" "gemm_big_kernel<3" gemm_big_multi "gemm_kernel" wgrad_x3_wide wgrad_x3_pipe semch_agg_fwd semch_agg_bwd attn_fwd attn_bwd; do
  echo "-- $k"; python scripts/pmc_kernel.py "$k" $F
done | tee $O/pmc_lds_summary.txt
