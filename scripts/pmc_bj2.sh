cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GEMM_TABLE_MAXM=4000 GAST_GEMM_BJ_ALL=1
i=0
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcbj2_$i -- python $R/scripts/gemm_table.py bf16x3 > /tmp/logbj2_$i.txt 2>&1
  for k in "gemm_bj_kernel<1, 2>" "gemm_bj_kernel<1, 1>"; do echo "-- $k"; python $R/scripts/pmc_kernel.py "$k" $(find /tmp/pmcbj2_$i -name "*counter_collection.csv") || tail -3 /tmp/logbj2_$i.txt; done
done
