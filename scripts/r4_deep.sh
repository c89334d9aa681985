#!/bin/bash
# gemm_big at prefetch distance 4 for the M = B*J stage (GAST_GEMM_BIG_DEEP=1): kernel tests with every big-kernel case forced deep, then the step A/B
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4j"; mkdir -p "$O"
GAST_GEMM_BIG_DEEP=1 GAST_GEMM_BIG_MIN_M=100000000 GAST_GEMM_BIG_DEEP_MIN_M=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "(gemm_big or big_ or test_gemm) and not optin" > "$O/tests_k.log" 2>&1
echo "kernel tests (all deep) rc=$? : $(tail -1 $O/tests_k.log)"
grep -E "^E |FAILED" "$O/tests_k.log" | head -10
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run base A=1
run deep GAST_GEMM_BIG_DEEP=1
run deep_k512 GAST_GEMM_BIG_DEEP=1 GAST_GEMM_BIG_DEEP_MAX_K=512
run deep_k1024 GAST_GEMM_BIG_DEEP=1 GAST_GEMM_BIG_DEEP_MAX_K=1024
run base2 A=1
