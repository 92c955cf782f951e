"""How long does the HOST need to enqueue one training step (no GPU back-pressure), and where does it go?

    python tests/studies/enqueue_cost.py [--prof 0|1] --out gpurun_out/enqueue_cost.json
After a device sync the step is enqueued and the wall time until Python returns is taken (the GPU consumes concurrently,
the queue never fills), then the GPU time of the same step.  If enqueue >= GPU time the step is launch-bound."""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import pacing_oracle as O  # noqa: E402
from pacingpseudo_amd._lib import lib  # noqa: E402
from pacingpseudo_amd.optim import FusedAdam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--prof', type=int, default=0)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--out', default='gpurun_out/enqueue_cost.json')
cli = ap.parse_args()
dev = torch.device('cuda', 0)
a = O.full_flags()
model = bench.build(a, dev)
opt = FusedAdam(model.parameters(), lr=a.lr, weight_decay=a.wd)
batch = {k: v.to(dev) for k, v in O.synthetic_batch(cli.batch, 256, 256, a.num_classes, seed=0).items() if k != 'label'}
model.train()
for _ in range(3):
    bench.train_iteration(model, opt, batch, a, 0)
torch.cuda.synchronize()
lib.pp_prof_enable(cli.prof)
rows = []
for _ in range(6):
    t0 = time.perf_counter()
    bench.train_iteration(model, opt, batch, a, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(dict(enqueue_ms=(t1 - t0) * 1e3, until_done_ms=(t2 - t0) * 1e3))
# back-to-back (what bench.py times)
t0 = time.perf_counter()
for _ in range(10):
    bench.train_iteration(model, opt, batch, a, 0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
bench.train_iteration(model, opt, batch, a, 0)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
res = dict(prof_events=cli.prof, per_step=rows, back_to_back=dict(enqueue_ms_per_step=(t1 - t0) * 100, total_ms_per_step=(t2 - t0) * 100),
           cprofile_top=s.getvalue().splitlines()[:60])
json.dump(res, open(cli.out, 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'cprofile_top'}, indent=1))
print('\n'.join(res['cprofile_top']))
