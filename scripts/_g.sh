for e in "GAST_GEMM_BIG_NARROW=0 GAST_GEMM_BIG_BWD_NI=4" "GAST_GEMM_BIG_NARROW=1 GAST_GEMM_BIG_BWD_NI=4" "GAST_GEMM_BIG_NARROW=0 GAST_GEMM_BIG_BWD_NI=2" "GAST_GEMM_BIG_NARROW=1 GAST_GEMM_BIG_BWD_NI=2" "GAST_GEMM_BIG_NARROW=0 GAST_GEMM_BIG_BWD_NI=4"; do
echo "== $e"; env $e timeout 200 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-parity 2>&1 | tail -1 | cut -c90-200
done
