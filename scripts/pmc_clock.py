"""Effective shader clock per kernel from one `rocprofv3 --pmc GRBM_GUI_ACTIVE` pass (MI355X_MICROARCH.md, 'DVFS give-back':
effective clock = GRBM_GUI_ACTIVE / 8 / dispatch wall time -- the counter is summed over the 8 XCDs; it reads high on dispatches
shorter than ~0.3 ms, so only dispatches of at least --min-us are used).

    python scripts/pmc_clock.py <dir with *counter_collection.csv> <out.json> [--min-us 150]"""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[sys.argv.index('--min-us') + 1]) if '--min-us' in sys.argv else 150.0
f = glob.glob(src + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
        continue
    dur_ns = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    if dur_ns < min_us * 1e3:
        continue
    acc[r['Kernel_Name'].split('(')[0].replace('void ', '').strip()].append((float(r['Counter_Value']) / 8.0 / dur_ns, dur_ns / 1e3))
res = {}
for k, v in acc.items():
    ghz = sorted(x for x, _ in v)
    res[k] = dict(dispatches=len(v), avg_us=round(sum(u for _, u in v) / len(v), 1), clock_ghz_median=round(ghz[len(ghz) // 2], 3),
                  clock_ghz_min=round(ghz[0], 3), clock_ghz_max=round(ghz[-1], 3))
res = dict(sorted(res.items(), key=lambda kv: -kv[1]['dispatches'] * kv[1]['avg_us']))
json.dump(dict(note='effective clock = GRBM_GUI_ACTIVE / 8 / (End - Start); peak 2.4 GHz; dispatches >= %g us only' % min_us, kernels=res),
          open(out, 'w'), indent=1)
for k, v in list(res.items())[:25]:
    print(f"{k[:70]:70s} n {v['dispatches']:4d} avg {v['avg_us']:7.1f} us  clock {v['clock_ghz_median']:.3f} GHz ({v['clock_ghz_min']:.2f}-{v['clock_ghz_max']:.2f})")
