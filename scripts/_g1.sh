mkdir -p gpurun_out/r02d
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/gputests.log 2>&1; echo "rc=$?" >> gpurun_out/r02d/gputests.log)
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02d/bench_x3.log 2>&1
tail -4 gpurun_out/r02d/gputests.log; tail -1 gpurun_out/r02d/bench_x3.log | cut -c1-400
