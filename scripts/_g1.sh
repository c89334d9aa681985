mkdir -p gpurun_out/r02a
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r02a/gputests.log)
timeout 600 python bench.py > gpurun_out/r02a/bench_default.log 2>&1
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r02a/bench_bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_x3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r02a/prof_x3.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/r02a/gputests.log; tail -2 gpurun_out/r02a/bench_default.log; tail -1 gpurun_out/r02a/bench_bf16.log
