"""Validation-Dice rows of the 16-bit STORAGE mode (round 4): per seed, the trajectory of `--storage fp16` (and of
`--storage fp16 --precision fp16`) against (a) the CPU-oracle (= reference path) trajectories of the same seed and (b) the fp32
HIP trajectory of the same seed, both taken from the committed rows of profiles/r03_dice_parity.json (same data, weights,
schedule: tests/studies/dice_study.py).  Dice = validation Dice averaged over the last five of 40 epochs.

    python tests/studies/dice_h16_summary.py --dir gpurun_out/dice_h16 --out profiles/r04_dice_storage_fp16.json"""
import argparse
import glob
import json
import math
import os
import re

from scipy import stats as st


def paired(diffs):
    n = len(diffs)
    mean = sum(diffs) / n
    sd = math.sqrt(sum((x - mean) ** 2 for x in diffs) / (n - 1)) if n > 1 else float('nan')
    half = st.t.ppf(0.975, n - 1) * sd / math.sqrt(n) if n > 1 else float('nan')
    return dict(n=n, mean_pt=round(100 * mean, 3), sd_pt=round(100 * sd, 3), ci95_half_width_pt=round(100 * half, 3),
                interval_pt=[round(100 * (mean - half), 3), round(100 * (mean + half), 3)], max_abs_pt=round(100 * max(abs(x) for x in diffs), 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default='gpurun_out/dice_h16')
    ap.add_argument('--base', default='profiles/r03_dice_parity.json')
    ap.add_argument('--out', default='profiles/r04_dice_storage_fp16.json')
    ap.add_argument('--dir256', default='', help='trajectories of scripts/dice_h16_256.sh (256-px phantoms, 10 epochs): added as `variants_256px`')
    a = ap.parse_args()
    base = json.load(open(a.base))
    out = None
    for size, d, hipvar in ((128, a.dir, 'final'), (256, a.dir256, 'final256')):
        if not d:
            continue
        part = summarise(base, a.base, size, d, hipvar)
        if out is None:
            out = part
        else:
            out['variants_256px'] = part['variants']
            out['what_256px'] = 'the same with 256-px phantoms (the benchmark geometry) and 10 epochs: mean of the last 5 of 10; 8 seeds'
            out['trajectories_256px'] = part['trajectories']
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.startswith('variants')}, indent=1))


def summarise(base, base_name, size, d, hipvar):
    class A:
        pass
    a = A()
    a.dir, a.base = d, base_name
    cols = base['trajectory_columns']
    cpu, hip32 = {}, {}
    for row in base['trajectories']:
        r = dict(zip(cols, row))
        if r['size'] != size or not r['finite']:
            continue
        if r['side'] == 'cpu':
            cpu.setdefault(r['seed'], []).append(r['last5_dice'])
        elif r['variant'] == hipvar:
            hip32[r['seed']] = r['last5_dice']
    cpu = {s: sum(v) / len(v) for s, v in cpu.items()}
    runs = {}
    for f in sorted(glob.glob(os.path.join(a.dir, 'r04_hip_*_s*.json'))):
        m = re.match(r'r04_hip_(h16x1|h16|f32)_s(\d+)\.json', os.path.basename(f))
        j = json.load(open(f))
        if not m or not j.get('done'):
            continue
        rows = j['rows']
        runs.setdefault(m.group(1), {})[int(m.group(2))] = dict(
            last5=sum(r['dice'] for r in rows[-5:]) / 5, final=rows[-1]['dice'], finite=all(math.isfinite(r['loss']) for r in rows))
    out = dict(what='validation Dice (avg over classes 1..K-1), mean of the last 5 of 40 epochs; 128-px phantoms, 200 train / 64 val, '
                    'batch 8, full flags (tests/studies/dice_study.py); h16 = --storage fp16 (activations and activation gradients '
                    'in HBM as fp16, loss scale 1024), h16x1 = the same with fp16 operands (--precision fp16): BASELINE config 5; '
                    'cpu = mean of the CPU-oracle runs of the seed, hip_fp32 = the fp32 HIP trajectory of the seed '
                    f'(both from {a.base})', variants={}, trajectories=[], trajectory_columns=['variant', 'seed', 'final_epoch_dice', 'last5_dice', 'finite'])
    f32 = runs.pop('f32', {})          # the fp32-storage path of the same binary (256-px study): pairs on every seed
    for v, per in sorted(runs.items()):
        seeds = sorted(s for s in per if s in cpu and s in hip32)
        d_cpu = [per[s]['last5'] - cpu[s] for s in seeds]
        d_hip = [per[s]['last5'] - hip32[s] for s in seeds]
        out['variants'][v] = dict(seeds=seeds, all_finite=all(per[s]['finite'] for s in per),
                                  mean_dice=round(sum(per[s]['last5'] for s in seeds) / len(seeds), 5),
                                  mean_dice_cpu=round(sum(cpu[s] for s in seeds) / len(seeds), 5),
                                  mean_dice_hip_fp32=round(sum(hip32[s] for s in seeds) / len(seeds), 5),
                                  minus_cpu=paired(d_cpu), minus_hip_fp32=paired(d_hip))
        both = sorted(s for s in per if s in f32)
        if len(both) > 1:
            out['variants'][v]['minus_fp32_storage_same_binary'] = dict(
                seeds=both, **paired([per[s]['last5'] - f32[s]['last5'] for s in both]),
                final_epoch=paired([per[s]['final'] - f32[s]['final'] for s in both]))
        for s in sorted(per):
            out['trajectories'].append([v, s, round(per[s]['final'], 5), round(per[s]['last5'], 5), int(per[s]['finite'])])
    for s in sorted(f32):
        out['trajectories'].append(['f32', s, round(f32[s]['final'], 5), round(f32[s]['last5'], 5), int(f32[s]['finite'])])
    return out


if __name__ == '__main__':
    main()
