#!/bin/bash
# round-3 GPU call 1: the whole GPU suite (all failures, not -x) + the weight-gradient product-count lever (accuracy and step time)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
rm -f gpurun_out/model_parity_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 --deselect tests/test_model_gpu.py::test_fp8_mixed_mode_configs4 > gpurun_out/r3a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -40 gpurun_out/r3a/tests.log
cp gpurun_out/model_parity_metrics.jsonl gpurun_out/r3a/metrics_suite.jsonl
for np in 3 2 1; do
  GAST_WGRAD_X3_PRODUCTS=$np timeout 300 python -m pytest tests/test_model_gpu.py -q -p no:cacheprovider -k "full_size_values and bf16x3 and 128-128-dilated" > gpurun_out/r3a/np${np}_test.log 2>&1
  GAST_WGRAD_X3_PRODUCTS=$np timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3a/bench_np$np.json 2> gpurun_out/r3a/bench_np$np.err
  python -c "import json;d=json.loads(open('gpurun_out/r3a/bench_np$np.json').read().strip().splitlines()[-1]);print('np',$np,d['ms_per_step'],d['value'],d.get('parity'))"
done
cp gpurun_out/model_parity_metrics.jsonl gpurun_out/r3a/metrics_all.jsonl
