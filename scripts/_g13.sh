mkdir -p gpurun_out/r02g
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02g/bench_x3.log 2>&1
tail -1 gpurun_out/r02g/bench_x3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms_per_step','value','eager_launch')}); print(d['roofline']); print(d['forward_only'])"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02g/trace -- python $R/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --steps 6 --warmup 2 > $R/gpurun_out/r02g/trace.log 2>&1
