timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('forward_only',{}).get('ms'))"
