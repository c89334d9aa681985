#!/bin/bash
# Round-6 evidence (run through gpurun): the driver's default bench line, the replayed step and the forward-only graph launch by launch
# (rocprofv3 --kernel-trace), per-kernel stats, the HBM-byte and MFMA-utilisation counter passes (separate --pmc runs), two more default
# lines for the median, the other configurations.  Everything lands in gpurun_out/r06_evidence; copy what is cited into profiles/.
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
O=$R/gpurun_out/r06_evidence; mkdir -p $O
if [ "$ONLY" != trace ] && [ "$ONLY" != prof ]; then
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json; echo
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --no-eager --no-twin > $O/bench_repeat_$i.json 2>/dev/null; done
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 8 --warmup 2 > $O/trace.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
T=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_step.py $T 3 > $O/step_summary.txt
python scripts/trace_timeline.py $T $O/step_timeline.txt
python scripts/trace_timeline.py $T $O/forward_timeline.txt fwd
python - "$T" > $O/forward_step_summary.txt <<'PY'
import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'expand_fwd_kernel' in r['Kernel_Name']]
# the forward-only graph: the LAST replays of the trace (no expand_bwd between two expand_fwd)
segs = []
for a, b in zip(marks[:-1], marks[1:]):
    if not any('expand_bwd_kernel' in rows[i]['Kernel_Name'] for i in range(a, b)):
        segs.append((a, b))
segs = sorted(sorted(segs[-6:], key=lambda ab: int(rows[ab[1]]['Start_Timestamp']) - int(rows[ab[0]]['Start_Timestamp']))[:3])      # (the shortest: no tracer stalls)
agg = collections.defaultdict(lambda: [0, 0]); busy = 0; wall = 0
for a, b in segs:
    wall += int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
    for r in rows[a:b]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'^void ', '', n)
        m = re.match(r'([\w:]+)', n); k = m.group(1) if m else n[:40]
        if 'at::native' in n: k = 'torch elementwise/fill'
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[k][0] += 1; agg[k][1] += d; busy += d
n = max(1, len(segs))
print('forward-only graph (train-mode forward incl. parameter packing): replays=%d kernels/replay=%.0f wall=%.1f us busy=%.1f us' % (n, sum(v[0] for v in agg.values()) / n, wall / 1e3 / n, busy / 1e3 / n))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('  %-40s n=%5.1f  us=%8.1f  avg=%7.1f us  %5.1f%%' % (k, c / n, t / 1e3 / n, t / 1e3 / c, 100.0 * t / busy))
PY
head -4 $O/forward_step_summary.txt
if [ "$ONLY" = trace ]; then head -12 $O/step_summary.txt; exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph --no-twin --no-f16 --no-stock-baseline > $O/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv") $O/pmc_hbm_bytes_bf16x3.json 3 > $O/pmc_summary.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timer --no-graph --no-twin --no-f16 --no-stock-baseline > $O/pmc_mfma.log 2>&1
cp $(find /tmp/pmc_mfma -name "*counter_collection.csv") $O/pmc_mfma_counters.csv
if [ "$ONLY" = prof ]; then ls $O; exit 0; fi
for cfg in cfg2 cfg3 cfg4 cfg243; do
  # (with the per-kernel timer -> roofline_by_kernel, and the CPU restatement's forward in the parity object; no CPU timing)
  timeout 900 python bench.py --config $cfg --no-cpu-baseline --cpu-parity --no-eager --no-stock-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  python -c "import json;d=json.loads(open('$O/bench_$cfg.json').read().strip().splitlines()[-1]);p=d['parity'];print('$cfg', d['ms_per_step'], d['value'], p.get('pass'), p.get('vs_fp32_hip',{}).get('max_abs'), (p.get('vs_cpu_reference_restatement') or {}).get('max_abs'), (d.get('roofline') or {}).get('frac'))"
done
ls $O
# one-rank run of the gradient exchange through RCCL (the 1-GPU box cannot measure scaling; this shows the collective path executes)
timeout 300 python bench.py --force-collective --no-cpu-baseline --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 10 --warmup 3 > $O/bench_force_collective.json 2> $O/bench_force_collective.err
tail -c 600 $O/bench_force_collective.json; echo
# LDS bank-conflict counters per kernel (its own --pmc run)
bash scripts/pmc_lds.sh $O > /dev/null 2>&1; head -12 $O/pmc_lds_summary.txt
# the M = B*J kernel (gemm_bj.hip) off / on every eligible shape, next to the default on the same box
bash scripts/ab_env.sh $O/ab_bj base= nobj=GAST_GEMM_BJ=0 allbj=GAST_GEMM_BJ_ALL=1 | tee $O/ab_bj.txt
# the 16-bit mode: the replayed f16 step per kernel, and its round-5 kernels (16-bit gemm_big / wgrad_wide) switched off on the same box
GAST_HIP_DTYPE=f16 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f16 -- python bench.py --no-cpu-baseline --no-parity --no-kernel-timer --no-eager --no-twin --no-f16 --no-stock-baseline --steps 8 --warmup 2 > $O/trace_f16.log 2>&1
python scripts/trace_step.py $(find /tmp/prof_f16 -name "*kernel_trace.csv" | head -1) 3 > $O/f16_step_summary.txt
bash scripts/ab_env.sh $O/ab_f16 "r4kernels=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0,GAST_WGRAD_H16_WIDE=0" "noimages=GAST_HIP_DTYPE=f16,GAST_H16_IMAGES=0" "default=GAST_HIP_DTYPE=f16" | tee $O/ab_f16.txt
