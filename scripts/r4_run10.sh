#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4m"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "semch_agg or deferred or (optin and AGG)" > "$O/tests_k.log" 2>&1
echo "kernel tests rc=$? : $(tail -1 $O/tests_k.log)"
GAST_TEST_H16=f16 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "semch_agg" > "$O/tests_k16.log" 2>&1
echo "kernel tests f16 rc=$? : $(tail -1 $O/tests_k16.log)"
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -m gpu -x > "$O/tests_m.log" 2>&1
echo "model+modules tests rc=$? : $(tail -1 $O/tests_m.log)"
grep -E "^E |FAILED" "$O/tests_k.log" "$O/tests_m.log" | head -10
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'), (d.get('variants') or {}).get('f16',{}).get('ms_per_step'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run new A=1
run old GAST_AGG_BWD_LDS=0
run new2 A=1
