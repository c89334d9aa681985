# SQ counters of the M = B*J GEMM kernel (gemm_bj.hip) on the step's small-M launches, every eligible shape on it (GAST_GEMM_BJ_ALL=1)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GEMM_TABLE_MAXM=4000 GAST_GEMM_BJ_ALL=1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcbj_$i -- python $R/scripts/gemm_table.py bf16x3 > /tmp/logbj_$i.txt 2>&1
  python $R/scripts/pmc_kernel.py "gemm_bj_kernel<1, 2>" $(find /tmp/pmcbj_$i -name "*counter_collection.csv") || tail -3 /tmp/logbj_$i.txt
done
