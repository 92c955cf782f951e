"""Find the first non-finite value of a HIP training trajectory (same data / schedule as dice_study.py) and report
which quantity went first: each loss, the memory bank, the gradient slab, the parameter slab, the BN running buffers.

    python tests/studies/nan_hunt.py --seed 2 --out gpurun_out/nan_hunt_s2.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pacing_oracle as O  # noqa: E402
from pacingpseudo_amd.data import SyntheticPhantoms  # noqa: E402
from tests.studies.dice_study import epoch_batches  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seed', type=int, default=2); ap.add_argument('--size', type=int, default=128)
ap.add_argument('--batch', type=int, default=8); ap.add_argument('--train', type=int, default=200)
ap.add_argument('--epochs', type=int, default=40); ap.add_argument('--wino', type=int, default=1)
ap.add_argument('--f16x3', type=int, default=1); ap.add_argument('--out', default='gpurun_out/nan_hunt.json')
a = ap.parse_args()

from pacingpseudo_amd import engine as E  # noqa: E402
E.WINO_ENABLED, E.F16X3_ENABLED = bool(a.wino), bool(a.f16x3)
from pacingpseudo_amd.optim import FusedAdam  # noqa: E402
from tests.test_gpu_step import build_model  # noqa: E402

args = O.full_flags(epoch=a.epochs)
sd = O.init_state(args, seed=a.seed)
model = build_model(args, {k: v.numpy() for k, v in sd.items()})
model.train()
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
tr = SyntheticPhantoms(a.train, args.num_classes, size=a.size, do_strong=True, train=True, seed=a.seed)
hist, step, found = [], 0, None


def snapshot(out, ep):
    bank = model.aux_path.memory_bank.detach()
    bufs = {n: b for n, b in model.named_buffers() if b.dtype.is_floating_point}
    rec = dict(step=step, epoch=ep, losses={k: float(v) for k, v in out.items() if k.startswith('loss')},
               bank_absmax=float(bank.abs().max()), bank_row_norms=[float(x) for x in bank.flatten(1).norm(dim=1)],
               grad_absmax=float(model.flat.grads.abs().max()), param_absmax=float(model.flat.params.abs().max()),
               logit_absmax=float(out['segmentation/logits'].abs().max()),
               aux_logit_absmax=float(out['logits_aux_cls'].abs().max()),
               bn_var_min=float(min(b.min() for n, b in bufs.items() if n.endswith('running_var'))),
               bn_var_max=float(max(b.max() for n, b in bufs.items() if n.endswith('running_var'))))
    bad = [k for k, v in rec['losses'].items() if not np.isfinite(v)]
    for k in ('bank_absmax', 'grad_absmax', 'param_absmax', 'logit_absmax', 'aux_logit_absmax'):
        if not np.isfinite(rec[k]):
            bad.append(k)
    rec['non_finite'] = bad
    return rec


for ep in range(a.epochs):
    lr = O.lr_at(args.lr_decay, ep, args.epoch, args.lr)
    w = O.loss_weights(args, ep)
    for g in opt.param_groups:
        g['lr'] = lr
    for b in epoch_batches(tr, a.batch, ep, a.seed, True):
        b = {k: v.cuda() for k, v in b.items() if k not in ('label', 'label_strong')}
        out = model(b, mode='train', step=ep)
        loss = sum(out[k] * wt for k, wt in w.items())
        opt.zero_grad(); loss.backward()
        step += 1
        rec = snapshot(out, ep)
        hist.append(rec)
        hist = hist[-6:]
        if rec['non_finite']:
            found = rec
            # which parameter gradients are non-finite
            found['bad_grads'] = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()][:20]
            found['scribble_pixels_sample0'] = [int(b['scribble'][0, c].sum()) for c in range(args.num_classes + 1)]
            found['bad_grads'] = None
            found['grad_finite'] = {n: bool(torch.isfinite(p.grad).all()) for n, p in model.named_parameters() if p.grad is not None}
            plan = model.engine.last_plan

            def fin(t):
                t = t.torch() if hasattr(t, 'torch') else t
                return dict(finite=bool(torch.isfinite(t).all()), n_bad=int((~torch.isfinite(t)).sum()), absmax=float(t[torch.isfinite(t)].abs().max()) if bool(torch.isfinite(t).any()) else None)
            bufs = {'dlogits': plan.dlogits, 'g_head': plan.g_head, 'logits': out['segmentation/logits'], 'logits_strong': out['segmentation/logits_strong']}
            for k in plan.dcat:
                bufs[f'dcat{k}'] = plan.dcat[k]
                bufs[f'cat{k}'] = plan.cat[k]
            for k in plan.g_low:
                bufs[f'g_low{k}'] = plan.g_low[k]
            for k in plan.dpooled:
                bufs[f'dpooled{k}'] = plan.dpooled[k]
            for k in ('feat', 'dfeat', 'dz', 'lo', 'dlo'):
                bufs['aux_' + k] = plan.aux[k]
            for n, t in plan.zbuf.items():
                bufs['z:' + n] = t
            for n, t in plan.coef.items():
                bufs['coef:' + n] = t
            for n, t in plan.amax.items():
                bufs['amax:' + n] = t
            found['buffers'] = {k: fin(v) for k, v in bufs.items()}
            found['amax_values'] = {n: float(t) for n, t in plan.amax.items()}
            break
        opt.step()
    if found:
        break
    model.eval()
    print('epoch', ep, 'ok', json.dumps(hist[-1]['losses']), flush=True)
json.dump(dict(config=vars(a), found=found, history=hist), open(a.out, 'w'), indent=1)
print(json.dumps(dict(found=found, history=hist), indent=1))
