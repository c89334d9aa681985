#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4g"; mkdir -p "$O"
GAST_HIP_LIB_EXPERIMENT=na3 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm_big or big_" > "$O/tests_k.log" 2>&1
echo "kernel tests (na3) rc=$? : $(tail -1 $O/tests_k.log)"
grep -E "^E |FAILED" "$O/tests_k.log" | head -10
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run na2_a GAST_HIP_LIB_EXPERIMENT=na2
run na3_a GAST_HIP_LIB_EXPERIMENT=na3
run na2_b GAST_HIP_LIB_EXPERIMENT=na2
run na3_b GAST_HIP_LIB_EXPERIMENT=na3
run na3_ni2 GAST_HIP_LIB_EXPERIMENT=na3 GAST_GEMM_BIG_NI=2
run na2_ni2 GAST_HIP_LIB_EXPERIMENT=na2 GAST_GEMM_BIG_NI=2
