#!/bin/bash
# round 3: SQ counters of the aggregation kernels inside the step (what bounds them: issue, memory wait, LDS?)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r3s; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmca_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph --no-twin > $O/log_$i.txt 2>&1
  for k in semch_agg_fwd semch_agg_bwd attn_bwd_wave bn_bwd_apply; do python scripts/pmc_kernel.py $k $(find /tmp/pmca_$i -name "*counter_collection.csv"); done
done | tee $O/pmc_agg.txt
