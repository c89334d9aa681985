cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in "X=1" "GAST_AGG_FWD_JSPLIT=2 GAST_AGG_BWD_JSPLIT=2" "GAST_AGG_FWD_JSPLIT=4 GAST_AGG_BWD_JSPLIT=4" "GAST_AGG_BWD_BLOCKS=2048" "GAST_AGG_BWD_BLOCKS=4096 GAST_AGG_FWD_JSPLIT=3"; do
  rm -rf /tmp/prof
  env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --steps 6 --warmup 2 > /tmp/log.txt 2>&1
  echo "== $e"; python $R/scripts/kstat.py $(find /tmp/prof -name "*kernel_stats.csv" | head -1) semch_agg
done
