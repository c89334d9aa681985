#!/bin/bash
# Same-box A/B of prebuilt library variants: "base" = gast_hip/libgast_hip.so, any other name = gast_hip/libgast_hip_<name>.so
# (GAST_HIP_LIB_EXPERIMENT).  Usage (through gpurun): bash scripts/ab_libs.sh OUT "tests -k expr" name1 name2 ...   (each timed twice, interleaved)
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; K="$2"; shift 2; mkdir -p "$O"
for v in "$@"; do
  if [ "$v" = base ]; then unset GAST_HIP_LIB_EXPERIMENT; else export GAST_HIP_LIB_EXPERIMENT=$v; fi
  if [ -n "$K" ]; then timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "$K" > "$O/tests_$v.log" 2>&1; echo "$v tests rc=$? $(tail -1 $O/tests_$v.log)"; fi
done
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset GAST_HIP_LIB_EXPERIMENT; else export GAST_HIP_LIB_EXPERIMENT=$v; fi
    timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_${v}_$rep.json" 2> "$O/bench_${v}_$rep.err"
    python -c "import json;d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1]);print('$v rep $rep:', d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" || tail -3 "$O/bench_${v}_$rep.err"
  done
done
