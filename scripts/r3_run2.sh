#!/bin/bash
# round-3 GPU call 2: re-run of the failed / new tests, default bench line (twin variant leg), launch-by-launch timelines (step and forward)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3b; mkdir -p $O
rm -f gpurun_out/model_parity_metrics.jsonl
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_inference_gpu.py tests/test_reference_caller.py "tests/test_model_gpu.py::test_golden" "tests/test_model_gpu.py::test_training_trajectory_matches_reference" "tests/test_model_gpu.py::test_module_graph_mode_matches_eager" -m gpu -q -p no:cacheprovider --timeout=600 -k "not bf16-" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|^E  |passed|failed" $O/tests.log | head -40
cp gpurun_out/model_parity_metrics.jsonl $O/metrics.jsonl
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d.get('variants'),d['parity']['pass'],d.get('forward_only'))"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --steps 6 --warmup 2 > $O/trace.log 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_step.py $T 3 > $O/step_summary.txt
python scripts/trace_timeline.py $T $O/step_timeline.txt
python scripts/trace_timeline.py $T $O/fwd_timeline.txt fwd
head -5 $O/step_summary.txt
