#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_streaming_gpu.py tests/test_inference_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "dropout_statistics or stream or module_graph or trajectory" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "^FAILED|passed|failed|^E   " $O/tests.log | head -20
timeout 300 python scripts/stream_bench.py > $O/stream_bench.txt 2>&1; tail -4 $O/stream_bench.txt
bash scripts/collect_evidence_r03.sh > $O/evidence.log 2>&1
tail -25 $O/evidence.log
