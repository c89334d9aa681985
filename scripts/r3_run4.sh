#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=gpurun_out/r3d; mkdir -p $O
bash scripts/ab_env.sh $O "base:GAST_X=0" "nomulti:GAST_GEMM_MULTI_SPLITK=0" "sk768:GAST_GEMM_SPLITK_BLOCKS=768" "sk1024:GAST_GEMM_SPLITK_BLOCKS=1024" \
   "bigsmall512:GAST_GEMM_BIG_SMALL_MIN_M=2048" "bigsmall1024:GAST_GEMM_BIG_SMALL_MIN_M=2048 GAST_GEMM_BIG_SMALL_MAX_K=1024" \
   "bigsmall1024_sk768:GAST_GEMM_BIG_SMALL_MIN_M=2048 GAST_GEMM_BIG_SMALL_MAX_K=1024 GAST_GEMM_SPLITK_BLOCKS=768"
GAST_GEMM_BIG_SMALL_MIN_M=2048 GAST_GEMM_BIG_SMALL_MAX_K=1024 timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "full_size_values and 128-128" > $O/tests_bigsmall.log 2>&1
tail -3 $O/tests_bigsmall.log
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --steps 6 --warmup 2 > $O/trace.log 2>&1
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_step.py $T 3 > $O/step_summary.txt
python scripts/trace_timeline.py $T $O/step_timeline.txt
head -3 $O/step_summary.txt
