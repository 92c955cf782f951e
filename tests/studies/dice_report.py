"""Fill the Dice paragraph of DESIGN.md from profiles/r02_dice_parity.json (tests/studies/dice_compare.py):
    python tests/studies/dice_report.py            # prints the paragraph;  --write replaces the marked block in DESIGN.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
j = json.load(open(os.path.join(ROOT, 'profiles', 'r02_dice_parity.json')))
p = j['pooled_last5']
d = p['hip_minus_cpu']
runs = p['runs_per_seed']
n_hip = sum(v.get('hip', 0) for v in runs.values())
n_cpu = sum(v.get('cpu', 0) for v in runs.values())
lines = [
    f"Result over {d['n']} seeds ({n_hip} HIP trajectories: the final binary, two earlier stages of this round, the Winograd-off /",
    "     split-fp16-off / fp32-direct corners and kernel-selection knobs that only change the summation order",
    f"     (`scripts/dice_variants.sh`); {n_cpu} CPU-reference trajectories with different thread counts, i.e. summation orders;",
    "     bit-identical trajectories counted once): mean validation Dice over the last five epochs, averaged per seed over the",
    "     runs of each side,",
    f"     **HIP {100 * p['mean_dice']['hip']:.2f} vs CPU {100 * p['mean_dice']['cpu']:.2f}: HIP − CPU = {d['mean_pt']:+.2f} pt, 95 % confidence interval ±{d['ci95_pt']:.2f} pt**",
    f"     (Student t over the per-seed differences, sd {d['sd_pt']:.2f} pt, largest single seed {d['max_abs_pt']:.2f} pt). Run-to-run standard",
    f"     deviation on one seed: HIP {p['run_to_run_sd_pt']['hip']:.2f} pt across kernel configurations, CPU {p['run_to_run_sd_pt']['cpu']:.2f} pt across thread counts.",
    "     Per kernel configuration against the first CPU run of each seed (noisier: one run against one run):",
]
for v, r in sorted(j['variants'].items()):
    m = r['mean_of_last_5_epochs']
    if m.get('ci95_pt') is None:
        continue
    lines.append(f"     `{v}` {m['mean_pt']:+.2f} ± {m['ci95_pt']:.2f} pt (n = {m['n']});")
lines[-1] = lines[-1].rstrip(';') + '.'
lines.append("     (`hip_shipped` = the binary of mid-round, five seeds only; its successors `hip_fusedbn` and `hip_final` run the same")
lines.append("     arithmetic -- the spread between these three rows is the noise of one-run-against-one-run comparisons.)")
text = '\n'.join(lines)
print(text)
if '--write' in sys.argv:
    path = os.path.join(ROOT, 'DESIGN.md')
    s = open(path).read()
    a, b = '     <!-- dice:begin -->\n', '     <!-- dice:end -->\n'
    if 'DICE_RESULTS_PLACEHOLDER' in s:
        s = s.replace('     DICE_RESULTS_PLACEHOLDER\n', a + '     ' + text + '\n' + b)
    else:
        i, k = s.index(a), s.index(b)
        s = s[:i] + a + '     ' + text + '\n' + s[k:]
    open(path, 'w').write(s)
