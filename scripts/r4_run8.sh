#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4h"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad or gemm_big or big_" > "$O/tests_k.log" 2>&1
echo "kernel tests rc=$? : $(tail -1 $O/tests_k.log)"
grep -E "^E |FAILED" "$O/tests_k.log" | head -10
GAST_TEST_H16=f16 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad and not x3" > "$O/tests_k16.log" 2>&1
echo "kernel tests f16 rc=$? : $(tail -1 $O/tests_k16.log)"
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'), (d.get('variants') or {}).get('f16'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run a A=1
run b A=1
export GAST_HIP_DTYPE=bf16x3
timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
GAST_HIP_DTYPE=bf16 timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
