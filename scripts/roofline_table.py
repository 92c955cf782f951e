"""Per-kernel roofline table of one profiled bench run, from the committed rocprofv3 summaries:
    python scripts/roofline_table.py r05 > profiles/r05_roofline_table.md
kernel_stats.csv (calls, average duration; two-stream steps AND the one-stream steps of the `single_stream` leg, so an
"average" here mixes both), hbm_traffic_per_launch.json (PMC FETCH/WRITE bytes per launch, kernels serialised by the PMC pass)
and sq_counters_per_kernel.json (MFMA-busy share of the busy CU cycles, alone on the chip).  Bandwidth = PMC bytes / average
duration, against 8 TB/s; the bound named is the larger of (bandwidth / 8 TB/s, MFMA-busy).
    python scripts/roofline_table.py r06 --one-stream > profiles/r06_one_stream_roofline_table.md
takes the durations from <tag>_one_stream_kernel_stats.csv instead (scripts/profile_one_stream.sh: the same bench command with the
weight gradients and the auxiliary path on the main stream, so every launch has the chip to itself -- the table to read a kernel's
own bandwidth from; the two-stream table stretches whatever shares the chip with the other stream)."""
import csv, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
one_stream = '--one-stream' in sys.argv[2:]
bn_mode = 'eval-mode BatchNorm (running statistics, the reference from epoch 1 on)' if 'evalbn' in tag else 'train-mode BatchNorm'
P = 'profiles/'
stats = list(csv.DictReader(open(f'{P}{tag}_one_stream_kernel_stats.csv' if one_stream else f'{P}{tag}_kernel_stats.csv')))
traffic = json.load(open(f'{P}{tag}_hbm_traffic_per_launch.json'))
sq = json.load(open(f'{P}{tag}_sq_counters_per_kernel.json'))
def key(n): return n.split('(')[0].replace('void ', '').strip()
tot = sum(float(r['TotalDurationNs']) for r in stats)
# steps of the traced run = launches of the optimizer's commit kernel / 2 (one per parameter segment: backbone, auxiliary path);
# profile_bench.sh runs 2 warm-up + 3 timed + 2 event-profiled + 3 one-stream steps
commits = sum(int(r['Calls']) for r in stats if 'optim_commit_kernel' in r['Name'])
steps = commits / 2.0 if commits else 10.0
print(f'# Per-kernel roofline table ({tag}{", ONE stream: every launch alone on the chip" if one_stream else ""}, 1x MI355X, batch 32, '
      f'256x256, full flags, {bn_mode})\n')
print(f'Kernel time per traced step: {tot / steps / 1e6:.2f} ms over {sum(int(r["Calls"]) for r in stats) / steps:.0f} launches '
      + ('(PP_WGRAD_STREAM=0 PP_AUX_SIDE=0: one stream, so this sum IS the step on the device; the shipped two-stream step is shorter).'
         if one_stream else '(sum over both streams: more than the step takes).')
      + '  HBM peak 8 TB/s; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES).\n')
print('| kernel | launches / step | avg us | ms / step | HBM MB / launch | TB/s | of 8 TB/s | MFMA-busy | bound |')
print('|---|---|---|---|---|---|---|---|---|')
for r in stats:
    k = key(r['Name'])
    ms = float(r['TotalDurationNs']) / steps / 1e6
    if ms < 0.02:
        continue
    avg = float(r['AverageNs']) / 1e3
    t = next((v for n, v in traffic.items() if isinstance(v, dict) and key(n) == k), None)
    s = next((v for n, v in sq.items() if key(n) == k), None)
    mb = t['hbm_bytes_per_launch'] / 1e6 if t else None
    tbs = (t['hbm_bytes_per_launch'] / (avg * 1e-6) / 1e12) if t else None
    mf = s.get('mfma_util') if s else None
    frac = tbs / 8.0 if tbs is not None else None
    bound = '-'
    if frac is not None or mf:
        bound = 'hbm' if (frac or 0) >= (mf or 0) else 'mfma'
    f = lambda x, d=2: '-' if x is None else f'{x:.{d}f}'
    print(f'| `{k[:64]}` | {int(r["Calls"]) / steps:.1f} | {avg:.0f} | {ms:.2f} | {f(mb, 0)} | {f(tbs)} | {f(frac)} | {f(mf)} | {bound} |')
