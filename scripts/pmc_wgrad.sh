cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GAST_HIP_DTYPE=bf16x3 GAST_MB_REPS=2 GAST_WGRAD_ORDER=1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$i -- python $R/scripts/wgrad_multi_bench.py s1 > /tmp/log_$i.txt 2>&1
  python $R/scripts/pmc_kernel.py wgrad_x3 $(find /tmp/pmc_$i -name "*counter_collection.csv") || tail -3 /tmp/log_$i.txt
done
