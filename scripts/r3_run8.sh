#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_caller.py tests/test_streaming_gpu.py tests/test_inference_gpu.py tests/test_train_tail.py -m gpu -q -p no:cacheprovider --timeout=600 -k "packed_operands or module_graph or trajectory or caller or stream or clip or flat_adam or data_parallel or flat_gradient" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|passed|failed|^E   " $O/tests.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-twin > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d.get('module_graph'), d.get('eager_launch'), d.get('forward_only'))"
timeout 300 python scripts/eval_forward_bench.py > $O/eval_forward.txt 2>&1; tail -5 $O/eval_forward.txt
