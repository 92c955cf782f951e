"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; FETCH_SIZE reports exactly half of the bytes
of a wide coalesced (16 B/lane) streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores."""
import collections, csv, glob, json, sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def load(d, counter):
    f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return agg


fe, wr = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
res = {}
for k in sorted(fe):
    n = len(fe[k])
    f_b = 2.0 * 1024.0 * sum(fe[k]) / n
    w_b = 1024.0 * sum(wr.get(k, [0.0])) / max(len(wr.get(k, [])), 1)
    res[k] = dict(launches=n, fetch_bytes_per_launch=f_b, write_bytes_per_launch=w_b, hbm_bytes_per_launch=f_b + w_b)
json.dump(res, open(out, 'w'), indent=1)
for k, v in res.items():
    print(f'{k[:60]:60s} n={v["launches"]:4d} fetch {v["fetch_bytes_per_launch"] / 1e6:9.1f} MB  write {v["write_bytes_per_launch"] / 1e6:9.1f} MB')
