#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r04_final; mkdir -p $O
S=$(date +%s)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
E=$(date +%s); echo "default bench wall: $((E-S)) s" | tee $O/bench_default.time
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['parity']['pass'], d['roofline']['frac'], d['roofline'].get('traffic_source','')[:40], d['forward_only']['ms'], d['variants']['f16']['ms_per_step'], d['variants']['f16']['parity']['pass'])"
