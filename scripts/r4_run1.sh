#!/bin/bash
# Round 4, GPU call 1: new kernel tests, whole-model parity subset, A/B of the lazy BatchNorm + weight-gradient chunk rule, timeline.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4a"; mkdir -p "$O"
timeout 600 python -m pytest tests/test_lazy_bn_gpu.py tests/test_kernels_gpu.py::test_prep_matches_contract tests/test_train_tail.py -q -m gpu -x > "$O/tests_new.log" 2>&1
echo "new tests rc=$? : $(tail -1 $O/tests_new.log)"
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "golden or trajectory or midsize or flat_gradient or module_graph or packed" > "$O/tests_model.log" 2>&1
echo "model tests rc=$? : $(tail -1 $O/tests_model.log)"
run() {  # name, env...
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run lazy GAST_BN_LAZY=1
run nolazy GAST_BN_LAZY=0
run lazy_chunk2176 GAST_WGRAD_MIN_CHUNK=2176
run lazy_chunk1088 GAST_WGRAD_MIN_CHUNK=1088
run lazy2 GAST_BN_LAZY=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python "$R/bench.py" --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --steps 6 --warmup 2 > "$O/prof_bench.log" 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python "$R/scripts/trace_step.py" "$T" 3 > "$O/step_summary.txt" 2>&1
python "$R/scripts/trace_timeline.py" "$T" "$O/step_timeline.txt" > /dev/null 2>&1
head -45 "$O/step_summary.txt"
