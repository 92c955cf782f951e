"""Validation-Dice parity after equal steps, FINAL BINARY ONLY on the HIP side.

    python tests/studies/dice_summary.py --hip <dir with r03_hip_final_s*.json> --cpu <dir> [<dir> ...] --out profiles/r03_dice_parity.json
    python tests/studies/dice_summary.py --from_compact profiles/r03_dice_parity.json          (recompute from the committed rows)

Pairs, per seed (= same data, initial weights, schedule), the ONE trajectory of the final HIP binary with the CPU-oracle
(= reference path) trajectories of that seed -- their mean when a seed has several (thread counts 2 / 4 / 8: different
summation orders of the same arithmetic) -- and states mean(HIP - CPU) of the validation Dice averaged over the last five
epochs with a 95 % Student-t interval.  Also written: one compact row per trajectory (the per-epoch JSON files are not
committed), the run-to-run spread of the CPU path on its own (seeds with >= 2 CPU runs) as the yardstick, and any other HIP
variant present (e.g. `fp16ops`: the mixed-precision mode)."""
import argparse
import glob
import json
import math
import os
import re

from scipy import stats as st


def load(dirs):
    runs = []
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, '*.json'))):
            m = re.match(r'r(\d+)_(cpu|hip)_(.+)_s(\d+)\.json', os.path.basename(f))
            if not m:
                continue
            j = json.load(open(f))
            if not j.get('done') or len(j['rows']) < 5:
                continue
            rows = j['rows']
            runs.append(dict(round=int(m.group(1)), side=m.group(2), variant=m.group(3), seed=int(m.group(4)),
                             final=rows[-1]['dice'], last5=sum(r['dice'] for r in rows[-5:]) / 5, epochs=len(rows),
                             finite=all(math.isfinite(r['loss']) for r in rows), threads=j['config'].get('threads'),
                             size=j['config'].get('size'), batch=j['config'].get('batch')))
    return runs


def load_compact(path):
    """The `trajectories` rows of an earlier summary (the per-epoch files are not committed): enough to recompute every figure."""
    j = json.load(open(path))
    cols = j['trajectory_columns']
    runs = []
    for row in j['trajectories']:
        r = dict(zip(cols, row))
        runs.append(dict(round=int(r['round'][1:]), side=r['side'], variant=r['variant'], seed=r['seed'], final=r['final_epoch_dice'],
                         last5=r['last5_dice'], finite=bool(r['finite']), threads=r['threads'], size=r['size'], batch=None, epochs=None))
    return runs


def ci(diffs):
    n = len(diffs)
    mean = sum(diffs) / n
    if n < 2:
        return dict(n=n, mean_pt=100 * mean)
    sd = math.sqrt(sum((x - mean) ** 2 for x in diffs) / (n - 1))
    h = st.t.ppf(0.975, n - 1) * sd / math.sqrt(n)
    se = sd / math.sqrt(n)
    # two one-sided tests of |true difference| < 0.5 pt: the larger of the two p-values (equivalence is shown when it is < 0.05)
    p_equiv = max(st.t.sf((0.005 - mean) / se, n - 1), st.t.sf((mean + 0.005) / se, n - 1)) if se > 0 else 0.0
    return dict(n=n, mean_pt=round(100 * mean, 3), sd_pt=round(100 * sd, 3), ci95_half_width_pt=round(100 * h, 3),
                interval_pt=[round(100 * (mean - h), 3), round(100 * (mean + h), 3)], max_abs_pt=round(100 * max(abs(x) for x in diffs), 3),
                p_equivalence_within_half_pt=round(float(p_equiv), 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hip', nargs='+', default=[])
    ap.add_argument('--cpu', nargs='+', default=[])
    ap.add_argument('--from_compact', default='', help='recompute from the trajectory rows of a committed summary instead of the per-epoch files')
    ap.add_argument('--final', default='final', help='variant name of the final binary')
    ap.add_argument('--min_hip_round', type=int, default=3, help='drop HIP trajectories of earlier rounds (their binaries are gone)')
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    hip = [r for r in load(a.hip) if r['side'] == 'hip' and r['round'] >= a.min_hip_round]
    cpu = [r for r in load(a.cpu) if r['side'] == 'cpu']
    if a.from_compact:                  # committed rows first; per-epoch files given besides them add (or replace) trajectories
        seen = {(r['side'], r['variant'], r['size'], r['seed'], r['threads']) for r in hip + cpu}
        for r in load_compact(a.from_compact):
            if (r['side'], r['variant'], r['size'], r['seed'], r['threads']) not in seen and not (r['side'] == 'hip' and r['round'] < a.min_hip_round):
                (hip if r['side'] == 'hip' else cpu).append(r)
    by_seed = {}
    for r in cpu:
        by_seed.setdefault((r['size'], r['seed']), []).append(r)
    res = dict(what='validation Dice (avg over classes 1..K-1), mean of the last 5 of 40 epochs; 128-px phantoms, 200 train / 64 val, '
                    'batch 8, full flags; HIP = one trajectory of the final binary per seed, CPU = mean of the CPU-oracle runs of that seed; '
                    'rows ending in _256px: the same with 256-px phantoms and 10 epochs (mean of the last 5 of 10)',
               variants={}, cpu_yardstick=None, trajectories=[])
    for variant in sorted({r['variant'] for r in hip}):
        for size in sorted({r['size'] for r in hip if r['variant'] == variant}):
            pairs = []
            for r in hip:
                if r['variant'] != variant or r['size'] != size or not r['finite']:
                    continue
                c = by_seed.get((size, r['seed']))
                if c:
                    pairs.append((r['seed'], r['last5'], sum(x['last5'] for x in c) / len(c), len(c)))
            pairs.sort()
            if pairs:
                key = variant if size == 128 else f'{variant}_{size}px'
                res['variants'][key] = dict(hip_minus_cpu=ci([h - c for _, h, c, _ in pairs]), seeds=[s for s, *_ in pairs],
                                            mean_dice_hip=round(sum(h for _, h, _, _ in pairs) / len(pairs), 5),
                                            mean_dice_cpu=round(sum(c for _, _, c, _ in pairs) / len(pairs), 5),
                                            cpu_runs_per_seed={str(s): n for s, _, _, n in pairs})
    # the CPU path against itself: pairs of CPU runs of one seed (different thread counts)
    d = []
    for (size, seed), c in sorted(by_seed.items()):
        if size == 128 and len(c) >= 2:
            d.append(c[0]['last5'] - c[1]['last5'])
    if d:
        res['cpu_yardstick'] = dict(what='CPU run A - CPU run B of the same seed (other thread count)', **ci(d))
    for r in sorted(hip + cpu, key=lambda x: (x['side'], x['variant'], x['size'] or 0, x['seed'])):
        res['trajectories'].append([f"r{r['round']:02d}", r['side'], r['variant'], r['size'], r['seed'], r['threads'], round(r['final'], 5),
                                    round(r['last5'], 5), int(r['finite'])])
    res['trajectory_columns'] = ['round', 'side', 'variant', 'size', 'seed', 'threads', 'final_epoch_dice', 'last5_dice', 'finite']
    print(json.dumps({k: v for k, v in res.items() if k != 'trajectories'}, indent=1))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(res, f, indent=0, separators=(',', ':'))
            f.write('\n')


if __name__ == '__main__':
    main()
