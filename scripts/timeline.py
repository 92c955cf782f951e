"""Two-stream timeline of one training step from a rocprofv3 --kernel-trace CSV (which kernel ran when, on which queue).

    python3 scripts/timeline.py <dir with *kernel_trace.csv> <out.tsv> [step index among the delimited steps, default 3]
Steps are delimited by the optimizer's commit kernel as in trace_gaps.py.  One line per dispatch: start (us from the first
dispatch of the step), duration (us), queue, kernel; and a summary of the stretches where only one queue was busy."""
import csv
import glob
import os
import sys

d, out = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:48], r.get('Queue_Id', '?')))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'optim_commit_kernel' in r[2]]
bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] > i + 2]
a, b = bounds[which], bounds[which + 1]
seg = rows[a + 1:b + 1]
t0 = seg[0][0]
queues = sorted({r[3] for r in seg}, key=lambda q: -sum(1 for r in seg if r[3] == q))
qn = {q: i for i, q in enumerate(queues)}
with open(out, 'w') as fo:
    fo.write(f'# step {which}: {len(seg)} dispatches, span {(seg[-1][1] - t0) / 1e3:.1f} us; queue 0 = most dispatches (main stream)\n')
    for s, e, k, q in seg:
        fo.write(f'{(s - t0) / 1e3:9.1f}\t{(e - s) / 1e3:8.1f}\tq{qn[q]}\t{k}\n')
    # busy time per queue and overlap
    ev = []
    for s, e, k, q in seg:
        ev.append((s, 1, qn[q])); ev.append((e, -1, qn[q]))
    ev.sort()
    act = {}
    last = ev[0][0]
    acc = {}
    for t, dlt, q in ev:
        key = tuple(sorted(k for k, v in act.items() if v > 0))
        acc[key] = acc.get(key, 0) + (t - last)
        last = t
        act[q] = act.get(q, 0) + dlt
    for key, v in sorted(acc.items()):
        fo.write(f'# queues busy {key}: {v / 1e6:.3f} ms\n')
print(open(out).read()[-600:])
