#!/bin/bash
# A/B of run-time switches with ONE binary on ONE box: bash scripts/ab_env.sh out_dir "name:VAR=val VAR2=val" ...
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; shift; mkdir -p "$O"
for v in "$@"; do
  name="${v%%:*}"; envs="${v#*:}"
  best=""
  for rep in 1 2; do
    env $envs timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --steps 30 --warmup 5 > "$O/bench_${name}_$rep.json" 2> "$O/bench_${name}_$rep.err"
    ms=$(python -c "import json;d=json.loads(open('$O/bench_${name}_$rep.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], '%.2e' % d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" 2>/dev/null)
    best="$best | $ms"
  done
  echo "$name [$envs]  ms/step, parity, fwd ms: $best"
done
