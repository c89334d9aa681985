"""Launch-by-launch timeline of the LAST replayed step of a rocprofv3 --kernel-trace CSV of a bench run (steps delimited by the
once-per-step expand_bwd kernel): index, start offset, duration, gap before, grid / block size, LDS, kernel name with its template
arguments.  With `fwd` as third argument the step is delimited by expand_fwd instead (forward-only graph replays).
Usage: python scripts/trace_timeline.py <kernel_trace.csv> [out.txt] [fwd]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
delim = 'expand_fwd_kernel' if (len(sys.argv) > 3 and sys.argv[3] == 'fwd') else 'expand_bwd_kernel'
marks = [i for i, r in enumerate(rows) if delim in r['Kernel_Name']]
if len(marks) < 2:
    print('could not delimit steps'); sys.exit(1)
# the shortest of the last four delimited steps (the tracer occasionally stalls a replay for milliseconds while it drains its buffers).
# A training step runs from the first kernel after an Adam launch to the next Adam launch inclusive; a forward-only replay from one
# expand_fwd to the next.
def span(ab):
    return int(rows[ab[1] - 1]['End_Timestamp']) - int(rows[ab[0]]['Start_Timestamp'])
if delim == 'expand_bwd_kernel':
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    pairs = [(a + 1, b + 1) for a, b in zip(adam[:-1], adam[1:])] if len(adam) >= 2 else list(zip(marks[:-1], marks[1:]))
    pairs = [ab for ab in pairs if any(delim in rows[i]['Kernel_Name'] for i in range(ab[0], ab[1]))]      # (whole training steps only)
else:
    pairs = [ab for ab in zip(marks[:-1], marks[1:]) if not any('expand_bwd_kernel' in rows[i]['Kernel_Name'] for i in range(ab[0], ab[1]))]
i0, i1 = min(pairs[-4:], key=span)
seg = rows[i0:i1]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('unsigned short', 'bf16').replace('float', 'f32')
    if 'at::native' in n: n = 'torch:' + (re.search(r'(\w+Functor|\w+_kernel)', n).group(1) if re.search(r'(\w+Functor|\w+_kernel)', n) else 'op')
    return n[:90]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
t0 = int(seg[0]['Start_Timestamp'])
prev = t0
tot = 0
for i, r in enumerate(seg):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    w = r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))
    try:
        blocks = int(g) // max(1, int(w))
    except Exception:
        blocks = -1
    tot += e - s
    out.write('%3d  t=%8.1f  dur=%7.1f  gap=%5.1f  blocks=%6d x %4s  lds=%6s  %s\n' % (i, (s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev) / 1e3, blocks, w,
                                                                                r.get('LDS_Block_Size', r.get('LDS_Block_Size_In_Bytes', '?')), short(r['Kernel_Name'])))
    prev = max(prev, e)
out.write('kernels=%d  span=%.1f us  busy=%.1f us\n' % (len(seg), (prev - t0) / 1e3, tot / 1e3))
