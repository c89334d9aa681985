#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4b"; mkdir -p "$O"
timeout 1500 python -m pytest tests -q -m gpu -x > "$O/tests_all.log" 2>&1
echo "gpu tests rc=$? : $(tail -1 $O/tests_all.log)"
run() {
  name="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --steps 30 --warmup 5 > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python -c "import json;d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null || { echo "$name FAILED"; tail -3 "$O/bench_$name.err"; }
}
run base A=1
run base2 A=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python "$R/bench.py" --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --steps 6 --warmup 2 > "$O/prof_bench.log" 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python "$R/scripts/trace_step.py" "$T" 3 > "$O/step_summary.txt" 2>&1
python "$R/scripts/trace_timeline.py" "$T" "$O/step_timeline.txt" > /dev/null 2>&1
head -12 "$O/step_summary.txt"
