#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4c"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_f16_gpu.py -q -m gpu -x > "$O/tests_f16.log" 2>&1
echo "f16 tests rc=$? : $(tail -1 $O/tests_f16.log)"
grep -E "Error|assert|FAILED" "$O/tests_f16.log" | head -20
timeout 400 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --steps 30 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs']); print(json.dumps(d['variants'].get('f16'), indent=0)[:1500])"
grep f16 "$R/gpurun_out/model_parity_metrics.jsonl" | tail -20
