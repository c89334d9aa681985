#!/bin/bash
# Same-box A/B of run-time switches: each argument is a name=ENV1=v1,ENV2=v2 spec ("base=" for the defaults); timed twice, interleaved.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O="$1"; shift; mkdir -p "$O"
for rep in 1 2; do
  for spec in "$@"; do
    name="${spec%%=*}"; envs="${spec#*=}"
    ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
      timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 30 --warmup 5 > "$O/bench_${name}_$rep.json" 2> "$O/bench_${name}_$rep.err" )
    python -c "import json;d=json.loads(open('$O/bench_${name}_$rep.json').read().strip().splitlines()[-1]);print('$name rep $rep:', d['ms_per_step'], d['parity']['vs_fp32_hip']['max_abs'], d.get('forward_only',{}).get('ms'))" || tail -3 "$O/bench_${name}_$rep.err"
  done
done
