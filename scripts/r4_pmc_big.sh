#!/bin/bash
# SQ counters of gemm_big over the step (bench.py eager, 2 steps): where a wave's cycles go
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r4pmc; mkdir -p $O
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph --no-twin --no-f16 --no-stock-baseline --no-eager > /tmp/logb_$i.txt 2>&1
  for k in "gemm_big_kernel<0, false, 4" "gemm_big_kernel<1, false, 4" "gemm_big_kernel<2, false, 2" "gemm_big_kernel<3, false, 2" "wgrad_x3_wide"; do
    echo "-- $k"; python scripts/pmc_kernel.py "$k" $(find /tmp/pmcb_$i -name "*counter_collection.csv") 2>/dev/null
  done
done | tee $O/pmc_big.txt
