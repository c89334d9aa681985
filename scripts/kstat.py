"""Print calls / average duration of the kernels whose name contains any of the given substrings (rocprofv3 kernel_stats.csv)."""
import csv, sys
path, subs = sys.argv[1], sys.argv[2:]
for r in csv.DictReader(open(path)):
    if any(s in r['Name'] for s in subs):
        print('%-60s calls %5s  avg %9.1f us  total %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
