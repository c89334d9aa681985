mkdir -p gpurun_out/r02f
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02f/pmc_fetch -- $B > $R/gpurun_out/r02f/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02f/pmc_write -- $B > $R/gpurun_out/r02f/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r02f/pmc_mfma -- $B > $R/gpurun_out/r02f/pmc_mfma.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f/trace -- python $R/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --steps 6 --warmup 2 > $R/gpurun_out/r02f/trace.log 2>&1
cd $R; ls gpurun_out/r02f/*/*/ | head -30; tail -2 gpurun_out/r02f/pmc_mfma.log | cut -c1-300
