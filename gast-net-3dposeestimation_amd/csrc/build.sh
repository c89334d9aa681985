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
HDRS="$HERE/common.h $HERE/gemm_big.h $HERE/bn_finalize.h $HERE/wgrad_common.h $HERE/../../include/gast_hip.h"
SRCS="gemm gemm_big gemm_bj wgrad wgrad_wide graph_ops norm_ops pack_ops optim_ops data_ops"
# Two storage flavours of the SAME sources (common.h): libgast_hip.so keeps GAST_BF16 tensors as bfloat16, libgast_hip_f16.so
# (-DGAST_H16_F16) as IEEE binary16 -- GAST_HIP_DTYPE=f16, the 16-bit mode that meets the north star's 1e-2 bound.  Same ABI.
build_flavour() {   # $1 = object directory, $2 = output library, $3 = extra flags
  local BDIR="$1" LIB="$2" FL="$FLAGS $3"
  mkdir -p "$BDIR"
  local pids=() names=()
  for f in $SRCS; do
    want="$( (echo "$FL"; $HIPCC --version | head -2; cat "$HERE/$f.hip" $HDRS) | sha256sum | cut -d' ' -f1)"
    have="$(cat "$BDIR/$f.sha" 2>/dev/null || true)"
    if [ ! -f "$BDIR/$f.o" ] || [ "$want" != "$have" ]; then
      rm -f "$BDIR/$f.sha"
      ( $HIPCC $FL -c "$HERE/$f.hip" -o "$BDIR/$f.o" && echo "$want" > "$BDIR/$f.sha" ) &
      pids+=($!)
      names+=($f)
    fi
  done
  for i in "${!pids[@]}"; do wait "${pids[$i]}" || { echo "compile failed: ${names[$i]}.hip ($LIB)" >&2; exit 1; }; done
  local OBJS=""
  for f in $SRCS; do OBJS="$OBJS $BDIR/$f.o"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$LIB" $OBJS
  echo "built $LIB (${#pids[@]} of $(echo $SRCS | wc -w) translation units recompiled)"
}
# (the two flavours compile side by side: a from-scratch build is bounded by ONE gemm_big.hip compile, ~2 min)
build_flavour "$HERE/build" "$OUT" "" &
P1=$!
P2=""
if [ -z "$GAST_SKIP_F16" ]; then      # (GAST_SKIP_F16=1: kernel-development builds that only need the bfloat16 flavour)
  build_flavour "$HERE/build/f16" "$HERE/../gast_hip/libgast_hip_f16.so" "-DGAST_H16_F16" &
  P2=$!
fi
wait $P1 || exit 1
if [ -n "$P2" ]; then wait $P2 || exit 1; fi
