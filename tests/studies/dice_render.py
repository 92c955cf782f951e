"""Renders the Dice paragraph of DESIGN.md section 4 from profiles/r03_dice_parity.json and splices it between the
`<!-- dice:begin -->` / `<!-- dice:end -->` markers:   python tests/studies/dice_render.py [--write]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(os.path.join(ROOT, 'profiles', 'r03_dice_parity.json')))


def row(key):
    v = d['variants'][key]
    c = v['hip_minus_cpu']
    return v, c


v, c = row('final')
lines = []
lines.append(f"     Result, final binary only (`profiles/r03_dice_parity.json`, written by `tests/studies/dice_summary.py`; one HIP")
lines.append(f"     trajectory per seed, paired with the mean of the CPU-oracle trajectories of the same seed — three thread counts for")
lines.append(f"     seeds 1–10, one for the others): **n = {c['n']} seeds, HIP {100 * v['mean_dice_hip']:.2f} vs CPU {100 * v['mean_dice_cpu']:.2f}: HIP − CPU = "
             f"{c['mean_pt']:+.2f} pt, 95 % confidence interval ±{c['ci95_half_width_pt']:.2f} pt** = [{c['interval_pt'][0]:+.2f}, {c['interval_pt'][1]:+.2f}]")
lines.append(f"     (Student t over the per-seed differences, sd {c['sd_pt']:.2f} pt, largest single seed {c['max_abs_pt']:.2f} pt; two one-sided tests of")
lines.append(f"     |true difference| < 0.5 pt: p = {c['p_equivalence_within_half_pt']:.3f}).")
y = d['cpu_yardstick']
lines.append(f"     Yardstick — the CPU path against itself, same seed, another thread count (= another summation order): "
             f"{y['mean_pt']:+.2f} ± {y['ci95_half_width_pt']:.2f} pt")
lines.append(f"     (n = {y['n']}, sd {y['sd_pt']:.2f} pt, largest {y['max_abs_pt']:.2f} pt).")
if 'fp16ops' in d['variants']:
    v2, c2 = row('fp16ops')
    lines.append(f"     Mixed-precision mode (`--precision fp16`, same seeds, same CPU reference): {c2['mean_pt']:+.2f} ± {c2['ci95_half_width_pt']:.2f} pt (n = {c2['n']}).")
k256 = [k for k in d['variants'] if k.endswith('_256px')]
for k in k256:
    v3, c3 = row(k)
    lines.append(f"     256-px phantoms, 10 epochs (mean of epochs 6–10, while the curves still climb — the per-seed spread is several points "
                 f"on BOTH sides): {c3['mean_pt']:+.2f} ± {c3['ci95_half_width_pt']:.2f} pt (n = {c3['n']}, sd {c3['sd_pt']:.2f} pt).")
text = '\n'.join(lines)
print(text)
if '--write' in sys.argv:
    p = os.path.join(ROOT, 'DESIGN.md')
    s = open(p).read()
    a = s.index('<!-- dice:begin -->') + len('<!-- dice:begin -->')
    b = s.index('<!-- dice:end -->')
    s = s[:a] + '\n' + text + '\n     ' + s[b:]
    open(p, 'w').write(s)
    print('spliced into DESIGN.md')
