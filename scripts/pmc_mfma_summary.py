"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE counter_collection CSV:
util = MFMA-busy cycles / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)   (profiles/README.md).  Usage: python scripts/pmc_mfma_summary.py <csv> [out.json]"""
import csv, sys, json, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([\w:]+)(<[^(]*>)?', n)
    key = (m.group(1) + (m.group(2) or '')) if m else n[:60]
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    did = (key, r.get('Dispatch_Id'))
    if did not in seen:
        seen.add(did)
        cnt[key] += 1
out = {}
for k, c in agg.items():
    gui = c.get('GRBM_GUI_ACTIVE', 0.0)
    if gui <= 0 or c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) <= 0:
        continue
    n = max(1, cnt[k])
    out[k] = dict(launches=n, mfma_busy_cycles_per_launch=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / n), gui_active_cycles_per_launch=round(gui / n),
                  mfma_util=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024), 4))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['mfma_busy_cycles_per_launch'] * kv[1]['launches'])[:12]:
    print('%-62s n=%4d util %.3f' % (k[:62], v['launches'], v['mfma_util']))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
