"""Pair the per-seed trajectories of dice_study.py (HIP vs CPU oracle, same seed = same data / weights / schedule) and
state the validation-Dice parity after equal steps:  mean over seeds of (Dice_HIP - Dice_CPU) at the END of training
(last epoch, and mean of the last 5 epochs) with a 95 % confidence interval (Student t over the paired differences).

    python tests/studies/dice_compare.py <dir of per-epoch files> --out profiles/r02_dice_parity.json

Round 2's tool (pools kernel variants); the per-epoch files it read are no longer committed -- round 3 uses dice_summary.py, whose
compact output keeps one row per trajectory."""
import argparse
import glob
import json
import math
import os
import re

T95 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262}   # two-sided, dof


def load(d):
    runs = {}
    for f in sorted(glob.glob(os.path.join(d, '*.json'))):
        m = re.match(r'r\d+_(cpu|hip)_(.+)_s(\d+)\.json', os.path.basename(f))
        if not m:
            continue
        j = json.load(open(f))
        if not j.get('done'):
            continue
        runs[(m.group(1) + '_' + m.group(2), int(m.group(3)))] = j
    return runs


def stats(diffs):
    n = len(diffs)
    mean = sum(diffs) / n
    if n < 2:
        return dict(n=n, mean_pt=100 * mean, ci95_pt=None)
    sd = math.sqrt(sum((x - mean) ** 2 for x in diffs) / (n - 1))
    return dict(n=n, mean_pt=100 * mean, sd_pt=100 * sd, ci95_pt=100 * T95.get(n - 1, 2.0) * sd / math.sqrt(n),
                max_abs_pt=100 * max(abs(x) for x in diffs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--ref', default='cpu_ref')
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    runs = load(a.dir)
    variants = sorted({v for v, _ in runs})
    res = dict(reference=a.ref, variants={}, per_seed={})
    for v in variants:
        if v == a.ref:
            continue
        seeds = sorted(s for (vv, s) in runs if vv == v and (a.ref, s) in runs)
        if not seeds:
            continue
        last, last5, worst_epoch = [], [], []
        for s in seeds:
            x, r = runs[(v, s)]['rows'], runs[(a.ref, s)]['rows']
            finite = all(math.isfinite(q['loss']) for q in x)
            n = min(len(x), len(r))
            last.append(x[n - 1]['dice'] - r[n - 1]['dice'])
            last5.append(sum(q['dice'] for q in x[n - 5:n]) / 5 - sum(q['dice'] for q in r[n - 5:n]) / 5)
            worst_epoch.append(max(abs(x[i]['dice'] - r[i]['dice']) for i in range(n)))
            res['per_seed'].setdefault(str(s), {})[v] = dict(final=x[n - 1]['dice'], last5=sum(q['dice'] for q in x[n - 5:n]) / 5,
                                                             finite_losses=finite)
            res['per_seed'][str(s)][a.ref] = dict(final=r[n - 1]['dice'], last5=sum(q['dice'] for q in r[n - 5:n]) / 5)
        res['variants'][v] = dict(seeds=seeds, final_epoch=stats(last), mean_of_last_5_epochs=stats(last5),
                                  largest_single_epoch_gap_pt=100 * max(worst_epoch))
    # ---- pooled statement: per seed, the mean over every HIP kernel configuration vs the mean over every CPU run
    # (numerically equivalent configurations of one path differ by chaos only; pooling them averages it out)
    pooled, seen, dropped = {}, set(), []
    for (v, s), j in sorted(runs.items()):
        n = len(j['rows'])
        if not all(math.isfinite(q['loss']) for q in j['rows']):
            continue
        # two kernel configurations that round identically give the SAME trajectory: count it once
        key = (s, v.split('_')[0], tuple(round(q['dice'], 12) for q in j['rows']))
        if key in seen:
            dropped.append(f'{v}_s{s}')
            continue
        seen.add(key)
        pooled.setdefault(s, {}).setdefault(v.split('_')[0], []).append(sum(q['dice'] for q in j['rows'][n - 5:n]) / 5)
    seeds = sorted(s for s, d in pooled.items() if 'hip' in d and 'cpu' in d)
    diffs = [sum(pooled[s]['hip']) / len(pooled[s]['hip']) - sum(pooled[s]['cpu']) / len(pooled[s]['cpu']) for s in seeds]

    def within(kind):
        dev = [x - sum(v) / len(v) for s in seeds for v in [pooled[s][kind]] if len(v) > 1 for x in v]
        dof = sum(len(pooled[s][kind]) - 1 for s in seeds if len(pooled[s][kind]) > 1)
        return 100 * math.sqrt(sum(d * d for d in dev) / dof) if dof else None
    if seeds:
        res['pooled_last5'] = dict(seeds=seeds, duplicate_trajectories_dropped=dropped, runs_per_seed={str(s): {k: len(v) for k, v in pooled[s].items()} for s in seeds},
                                   hip_minus_cpu=stats(diffs),
                                   run_to_run_sd_pt=dict(hip=within('hip'), cpu=within('cpu')),
                                   mean_dice=dict(hip=sum(sum(pooled[s]['hip']) / len(pooled[s]['hip']) for s in seeds) / len(seeds),
                                                  cpu=sum(sum(pooled[s]['cpu']) / len(pooled[s]['cpu']) for s in seeds) / len(seeds)))
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
