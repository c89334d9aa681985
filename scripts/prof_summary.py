"""Summarise a rocprofv3 --kernel-trace CSV: per kernel totals, and the per-launch sequence of the last launches."""
import csv, sys, collections, re
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void ', '', n)
    return n[:70]
agg = collections.OrderedDict()
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name'])
    a = agg.setdefault(k, [0, 0.0, 1e18, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print('%-72s %7s %10s %8s %8s %8s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-72s %7d %10.1f %8.1f %8.1f %8.1f %6.1f' % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
if len(sys.argv) > 2:
    pat = sys.argv[2]
    sel = [r for r in rows if pat in r['Kernel_Name']]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    print('--- last %d launches matching %s' % (n, pat))
    for r in sel[-n:]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print('%-50s grid %-8s wg %-5s lds %-6s vgpr %-4s %8.1f us' % (short(r['Kernel_Name'])[:50], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), d))
