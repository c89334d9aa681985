#!/bin/bash
# final state of the round: the driver's default line (timed) and the step trace again, after the aggregation-backward change
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r04_final; mkdir -p $O
/usr/bin/time -v -o $O/bench_default.time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
grep -E "Elapsed" $O/bench_default.time
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['parity']['pass'], d['roofline']['frac'], d['roofline'].get('traffic_source','')[:40], d['forward_only']['ms'], d['variants']['f16']['ms_per_step'], d['variants']['f16']['parity']['pass'])"
ONLY=trace O_OVERRIDE=$O bash scripts/collect_evidence_r04.sh > $O/trace_run.log 2>&1
cp $R/gpurun_out/r04_evidence/step_summary.txt $R/gpurun_out/r04_evidence/step_timeline.txt $R/gpurun_out/r04_evidence/forward_step_summary.txt $R/gpurun_out/r04_evidence/forward_timeline.txt $R/gpurun_out/r04_evidence/kernel_stats.csv $O/ 2>/dev/null
head -12 $O/step_summary.txt
