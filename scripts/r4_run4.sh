#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4d"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "second_output or gemm_big_x3" > "$O/tests_k.log" 2>&1
echo "kernel tests rc=$? : $(tail -1 $O/tests_k.log)"
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "golden or midsize or full_size_values or dropout_gradients" > "$O/tests_m.log" 2>&1
echo "model tests rc=$? : $(tail -1 $O/tests_m.log)"
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run fused A=1
run unfused GAST_FUSE_MASK=0
run fused2 A=1
run ni2 GAST_GEMM_BIG_NI=2
run ni4 GAST_GEMM_BIG_NI=4
run bwdni4 GAST_GEMM_BIG_BWD_NI=4
