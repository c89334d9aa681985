#!/bin/bash
# A/B of compile-time kernel variants ON THE GPU BOX (hipcc is in the image): for each "name:flags" argument rebuild libgast_hip.so with
# EXTRA_FLAGS=flags, run the kernel tests of the touched kernels, then time the default step.  Usage (through gpurun):
#   bash scripts/ab_variants.sh out_dir "base:" "prio:-DGAST_MFMA_PRIO" ...
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; shift; mkdir -p "$O"
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  EXTRA_FLAGS="$flags" bash gast-net-3dposeestimation_amd/csrc/build.sh > "$O/build_$name.log" 2>&1 || { echo "$name: BUILD FAILED"; tail -5 "$O/build_$name.log"; continue; }
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "${AB_TESTS:-gemm or wgrad}" > "$O/tests_$name.log" 2>&1
  trc=$?
  best=""
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --steps 30 --warmup 5 > "$O/bench_${name}_$rep.json" 2> "$O/bench_${name}_$rep.err"
    ms=$(python -c "import json;d=json.loads(open('$O/bench_${name}_$rep.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null)
    best="$best | $ms"
  done
  echo "$name [$flags] tests_rc=$trc  ms/step, parity, fwd ms: $best"
done
