#!/bin/bash
# Order / state screen of the kernel suite on the GPU box (through gpurun): tests/test_kernels_gpu.py in N shuffled orders, every
# uninitialised allocation and the split-K workspace poisoned with NaN (tests/conftest.py: --gast-shuffle, GAST_TEST_POISON), in the
# bfloat16 flavour (the whole file) and in the binary16 flavour (the selection tests/test_f16_gpu.py runs as a child).  A bit-equality
# miss prints which elements differ (tests/test_kernels_gpu.py::_assert_bit_equal).  Usage: bash scripts/stress_kernel_suite.sh OUT [N]
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; N="${2:-6}"; mkdir -p "$O"
fails=0
for seed in $(seq 1 "$N"); do
  GAST_TEST_POISON=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --gast-shuffle "$seed" \
      -k "not optin" > "$O/bf16_seed$seed.log" 2>&1
  rc1=$?
  GAST_TEST_POISON=1 GAST_TEST_H16=f16 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider --gast-shuffle "$seed" \
      -k "(bf16 or dt1 or dtype1 or float16 or out_f32) and not optin and not fp8 and not x3" > "$O/f16_seed$seed.log" 2>&1
  rc2=$?
  echo "seed $seed: bf16 flavour rc=$rc1 ($(tail -1 "$O/bf16_seed$seed.log")) | f16 flavour rc=$rc2 ($(tail -1 "$O/f16_seed$seed.log"))"
  if [ $rc1 -ne 0 ] || [ $rc2 -ne 0 ]; then fails=$((fails + 1)); grep -h "^FAILED\|AssertionError" "$O/bf16_seed$seed.log" "$O/f16_seed$seed.log" | head -8; fi
done
echo "stress: $fails of $N shuffled orders had a failure"
