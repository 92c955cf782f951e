"""Idle time between kernels from a rocprofv3 --kernel-trace CSV: how launch-bound is the step?

    python3 scripts/trace_gaps.py <dir with *kernel_trace.csv> <out.json> [adam_kernel]
The steps are delimited by the optimizer kernel (one launch pair per step); reports, for the steps after the first two,
span, busy time (union of kernel intervals), idle time and the idle-gap histogram."""
import csv
import glob
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'optim_commit_kernel'      # (round 5: one commit kernel behind each optimizer launch)
f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                 (int(r.get('Grid_Size_X', 0) or 0), int(r.get('Grid_Size_Y', 0) or 0), int(r.get('Grid_Size_Z', 0) or 0)),
                 int(r.get('Workgroup_Size_X', 0) or 0), int(r.get('LDS_Block_Size', 0) or 0), int(r.get('VGPR_Count', 0) or 0)))
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
if not ends:
    marker = 'adam_kernel'
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
# the marker may launch several times per step (one per parameter slab, each followed by at most one other optimizer launch):
# a step boundary = last marker of a run
bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] > i + 2]
steps = []
for a, b in zip(bounds[:-1], bounds[1:]):
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, *_ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    hist = {'<2us': 0, '2-5us': 0, '5-10us': 0, '10-50us': 0, '>50us': 0}
    for g in gaps:
        k = '<2us' if g < 2000 else '2-5us' if g < 5000 else '5-10us' if g < 10000 else '10-50us' if g < 50000 else '>50us'
        hist[k] += 1
    steps.append(dict(kernels=len(seg), span_ms=span / 1e6, busy_ms=busy / 1e6, idle_ms=(span - busy) / 1e6,
                      idle_gaps=len(gaps), gap_hist=hist, largest_gaps_us=[round(g / 1e3, 1) for g in sorted(gaps)[-5:]]))
# every dispatch of the last step, in launch order: kernel, grid (work-items), block, LDS bytes, VGPRs, microseconds
a, b = bounds[-2], bounds[-1]
last = [dict(kernel=r[2].split('(')[0][:60], grid=list(r[3]), block=r[4], lds=r[5], vgpr=r[6], us=round((r[1] - r[0]) / 1e3, 1))
        for r in rows[a + 1:b + 1]]
res = dict(trace=os.path.basename(f), steps=steps[-4:], last_step_dispatches=last)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(dict(steps=res['steps']), indent=1))
