#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4e"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "fused_parameter_packing or packed_operands or training_forward_always or golden" > "$O/tests_m.log" 2>&1
echo "model tests rc=$? : $(tail -1 $O/tests_m.log)"
grep -E "^E |FAILED" "$O/tests_m.log" | head -10
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run fused A=1
run unfused GAST_PACK_FUSED=0
run fused2 A=1
run unfused2 GAST_PACK_FUSED=0
