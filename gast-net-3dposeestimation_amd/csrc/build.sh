#!/bin/bash
# Build libgast_hip.so (gfx950 only) in-tree, next to the ctypes binding.
# Incremental by CONTENT, not by mtime: an object is reused only if the SHA-256 of its source, every header it can include and the
# compiler flags equals the stamp written when it was built (a stale or foreign build/*.o is never silently linked).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../gast_hip/libgast_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment"
# ABLATION=1: profiling build of gemm_big.hip with the GAST_GEMM_BIG_ABLATE run-time switches compiled in (scripts/gemm_big_ablate.py)
if [ -n "$ABLATION" ]; then FLAGS="$FLAGS -DGAST_GEMM_BIG_ABLATION"; fi
# EXTRA_FLAGS="-DGAST_..." : experiment switches of single kernels (scripts/ab_variants.sh builds and times several variants on the GPU box)
if [ -n "$EXTRA_FLAGS" ]; then FLAGS="$FLAGS $EXTRA_FLAGS"; fi
mkdir -p "$HERE/build"
HDRS="$HERE/common.h $HERE/gemm_big.h $HERE/../../include/gast_hip.h"
SRCS="gemm gemm_big wgrad graph_ops norm_ops pack_ops optim_ops data_ops"
pids=()
names=()
for f in $SRCS; do
  want="$( (echo "$FLAGS"; $HIPCC --version | head -2; cat "$HERE/$f.hip" $HDRS) | sha256sum | cut -d' ' -f1)"
  have="$(cat "$HERE/build/$f.sha" 2>/dev/null || true)"
  if [ ! -f "$HERE/build/$f.o" ] || [ "$want" != "$have" ]; then
    rm -f "$HERE/build/$f.sha"
    ( $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/build/$f.o" && echo "$want" > "$HERE/build/$f.sha" ) &
    pids+=($!)
    names+=($f)
  fi
done
for i in "${!pids[@]}"; do wait "${pids[$i]}" || { echo "compile failed: ${names[$i]}.hip" >&2; exit 1; }; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $HERE/build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT (${#pids[@]} of $(echo $SRCS | wc -w) translation units recompiled)"
