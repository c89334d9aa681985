cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_evidence; mkdir -p $O
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --steps 6 --warmup 2 > $O/trace.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python $R/scripts/trace_step.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) 3 > $O/step_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph > $O/pmc_$c.log 2>&1
done
python $R/scripts/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv") $O/pmc_hbm_bytes.json 3 > $O/pmc_summary.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph > $O/pmc_mfma.log 2>&1
cp $(find /tmp/pmc_mfma -name "*counter_collection.csv") $O/pmc_mfma_counters.csv
ls -la $O
