cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02i
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --steps 6 --warmup 2 > $R/gpurun_out/r02i/bench.log 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/r02i/kernel_stats.csv
python $R/scripts/trace_step.py $T 3 > $R/gpurun_out/r02i/step_summary.txt
head -${1:-16} $R/gpurun_out/r02i/step_summary.txt
