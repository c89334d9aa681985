"""Sum rocprofv3 --pmc counter CSVs per kernel and counter.  Usage: python scripts/pmc_kernel.py <substr> <csv> [<csv> ...]"""
import csv, sys, collections
sub = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sub in r['Kernel_Name']:
            a = agg[r['Counter_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
for k in sorted(agg):
    n, v = agg[k]
    print('%-34s launches %4d  per launch %16.1f' % (k, n, v / n))
