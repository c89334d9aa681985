"""Per-step picture from a rocprofv3 --kernel-trace CSV of a bench run: takes the LAST full step (delimited by the
once-per-step expand_bwd kernel), prints wall span, summed kernel time, idle gaps and the per-kernel totals.
Usage: python scripts/trace_step.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    base = m.group(1) if m else n[:40]
    if 'multi_tensor_apply' in n: base = 'torch multi_tensor_apply (Adam)'
    elif 'FillFunctor' in n: base = 'torch fill'
    elif 'at::native' in n: base = 'torch elementwise/reduce'
    if 'gemm_kernel' in n or 'splitk_finish' in n:
        base += '<f32out>' if 'unsigned short, float' in n else ''
    return base
# steps are delimited by a kernel that runs exactly once per step
fills = [i for i, r in enumerate(rows) if 'expand_bwd_kernel' in r['Kernel_Name']]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(fills) < nlast + 1:
    print('could not delimit steps (fills=%d)' % len(fills)); sys.exit(1)
# the nlast steps with the shortest wall span among the last nlast + 3 (the tracer occasionally stalls a replay for milliseconds while
# it drains its buffers: such a step says nothing about the kernels)
cand = [(fills[i], fills[i + 1]) for i in range(max(0, len(fills) - 1 - (nlast + 3)), len(fills) - 1)]
cand.sort(key=lambda ab: int(rows[ab[1]]['Start_Timestamp']) - int(rows[ab[0]]['Start_Timestamp']))
chosen = sorted(cand[:nlast])
seg, wall, gaps, gap_list = [], 0, 0, []
for i0, i1 in chosen:
    part = rows[i0:i1]
    seg += part
    wall += int(rows[i1]['Start_Timestamp']) - int(part[0]['Start_Timestamp'])
    prev_end = int(part[0]['End_Timestamp'])
    for r in part[1:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > prev_end:
            gaps += s - prev_end
            gap_list.append((s - prev_end, short(r['Kernel_Name'])))
        prev_end = max(prev_end, e)
nlast = len(chosen)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print('steps=%d (shortest of the last %d) kernels/step=%.0f wall/step=%.1f us  busy/step=%.1f us  idle gaps/step=%.1f us' % (
    nlast, len(cand), len(seg) / nlast, wall / 1e3 / nlast, busy / 1e3 / nlast, gaps / 1e3 / nlast))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = short(r['Kernel_Name'])
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('  %-46s n/step=%6.1f  us/step=%8.1f  avg=%7.1f us  %5.1f%%' % (k, n / nlast, t / 1e3 / nlast, t / 1e3 / n, 100.0 * t / busy))
gagg = collections.defaultdict(lambda: [0, 0])
for g, k in gap_list:
    gagg[k][0] += 1; gagg[k][1] += g
print('idle before kernel (top):')
for k, (n, t) in sorted(gagg.items(), key=lambda kv: -kv[1][1])[:8]:
    print('  %-46s n/step=%6.1f  us/step=%8.1f' % (k, n / nlast, t / 1e3 / nlast))
