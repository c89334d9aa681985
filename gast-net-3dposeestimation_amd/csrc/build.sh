#!/bin/bash
# Build libgast_hip.so (gfx950 only) in-tree, next to the ctypes binding.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../gast_hip/libgast_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment"
# ABLATION=1: profiling build of gemm_big.hip with the GAST_GEMM_BIG_ABLATE run-time switches compiled in (scripts/gemm_big_ablate.py)
if [ -n "$ABLATION" ]; then FLAGS="$FLAGS -DGAST_GEMM_BIG_ABLATION"; touch "$HERE/gemm_big.hip"; fi
mkdir -p "$HERE/build"
pids=()
for f in gemm gemm_big wgrad graph_ops norm_ops pack_ops optim_ops data_ops; do
  if [ ! -f "$HERE/build/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/build/$f.o" ] || [ "$HERE/common.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/gemm_big.h" -nt "$HERE/build/$f.o" ] || [ "$HERE/../../include/gast_hip.h" -nt "$HERE/build/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/build/gemm.o" "$HERE/build/gemm_big.o" "$HERE/build/wgrad.o" "$HERE/build/graph_ops.o" "$HERE/build/norm_ops.o" "$HERE/build/pack_ops.o" "$HERE/build/optim_ops.o" "$HERE/build/data_ops.o"
echo "built $OUT"
