#!/bin/bash
# 256 x 256-tile bf16x3 weight gradient (wgrad_wide.hip) against the 128 x 128 pipelined kernel: kernel tests, then the three stage job sets
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$R/gpurun_out/r4w"; mkdir -p "$O"
GAST_WGRAD_X3_TILE=256 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad" > "$O/tests_k.log" 2>&1
echo "kernel tests (wide) rc=$? : $(tail -1 $O/tests_k.log)"
grep -E "^E |FAILED" "$O/tests_k.log" | head -10
export GAST_HIP_DTYPE=bf16x3
echo "== narrow"; timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
for nb in 256 512; do
echo "== wide, $nb blocks"; GAST_WGRAD_X3_TILE=256 GAST_WGRAD_BLOCKS_WIDE=$nb timeout 300 python scripts/wgrad_multi_bench.py s0 s1 s2 2>&1 | tail -3
done
