"""Print per-tensor relative errors of one HIP step against a golden case (debug aid, GPU box)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _golden as G
from tests.test_gpu_step import build_model, iteration
from pacingpseudo_amd.optim import FusedAdam
from pacingpseudo_amd.utils import poly_lr_decay

name = sys.argv[1] if len(sys.argv) > 1 else 'full_seq'
d = G.load(name); args = G.case_args(name); epochs = G.CASES[name][1]
model = build_model(args, G.sub(d, 'init/'))
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
opt, lr = poly_lr_decay(opt, epochs[0], args.epoch, args.lr)
rec, grads = iteration(model, opt, G.batch_of(d, 0), args, epochs[0])
for k, v in G.sub(d, 'step0/out/').items():
    print(f'out  {G.rel_err(rec[k].double().cpu().numpy(), v):.3e}  {k}')
rows = []
from oracle import pacing_oracle as O
sd = G.to_state(G.sub(d, 'init/'))
_, og, _ = O.train_step(sd, G.batch_of(d, 0), epochs[0], args, True)
ref_grads = {k: v.numpy() for k, v in og.items() if v is not None}
for k, v in ref_grads.items():
    if grads.get(k) is None:
        print('MISSING', k); continue
    rows.append((G.rel_err(grads[k].double().cpu().numpy(), v), float(np.max(np.abs(v))), k))
for e, m, k in rows:
    print(f'grad {e:.3e}  max|ref| {m:.3e}  {k}')
