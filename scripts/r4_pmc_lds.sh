#!/bin/bash
# LDS counters of the weight-gradient kernels on the C = 256 stage's job set (wide by default; WIDE_TILE=128 for the narrow kernel)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GAST_HIP_DTYPE=bf16x3 GAST_MB_REPS=2 GAST_WGRAD_X3_TILE=${WIDE_TILE:-256}
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_l -- python $R/scripts/wgrad_multi_bench.py s1 > /tmp/log_l.txt 2>&1
python $R/scripts/pmc_kernel.py wgrad_x3 $(find /tmp/pmc_l -name "*counter_collection.csv")
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_w -- python $R/scripts/wgrad_multi_bench.py s1 > /tmp/log_w.txt 2>&1
python $R/scripts/pmc_kernel.py wgrad_x3 $(find /tmp/pmc_w -name "*counter_collection.csv")
