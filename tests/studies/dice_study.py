"""Validation-Dice parity study: ONE training trajectory per invocation, on the CPU oracle (= the reference path) or on
the HIP path, over the SAME deterministic phantom data, initial weights and schedule (train_chaos.py:243-428 semantics:
per-epoch poly LR, loss ramp-ups, model.eval() after epoch 0 and never back, validation Dice per epoch = avg over
classes 1..K-1 of the per-class means over non-NaN samples, utils/metrics.py:7-34 + train_chaos.py:388-395).

    python tests/studies/dice_study.py --backend cpu --seed 3 --threads 4 --out gpurun_out/dice_cpu/r03_cpu_ref_s3.json
    python tests/studies/dice_study.py --backend hip --seed 3 --out gpurun_out/dice/hip_s3.json
    python tests/studies/dice_study.py --backend cpu --seed 3 --noise 1e-5 ...     (conv outputs perturbed: yardstick)

The CPU trajectories need no GPU (they are run in the build container, hours of host time); the HIP trajectories run
on the MI355X box.  `dice_compare.py` pairs them by seed.  Both backends score their logits with the SAME numpy Dice
(oracle.compute_dice), so the metric code cannot contribute to a difference."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pacing_oracle as O  # noqa: E402
from pacingpseudo_amd.data import SyntheticPhantoms  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=['cpu', 'hip'], required=True)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--train', type=int, default=200)
    ap.add_argument('--val', type=int, default=64)
    ap.add_argument('--epochs', type=int, default=40, help='epochs run = --epoch of the schedule')
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--noise', type=float, default=0.0, help='cpu only: relative gaussian noise on every conv output')
    ap.add_argument('--noise_seed', type=int, default=1234)
    ap.add_argument('--wino', type=int, default=1)
    ap.add_argument('--f16x3', type=int, default=1)
    ap.add_argument('--products', type=int, default=3, choices=[1, 3], help='hip only: 1 = the fp16-operand mixed-precision mode')
    ap.add_argument('--storage', default='fp32', choices=['fp32', 'fp16', 'bf16'], help='hip only: fp16 / bf16 = the 16-bit storage modes (training plans)')
    ap.add_argument('--snap', default='', help='comma list of global steps at which to record the parameter checksum')
    ap.add_argument('--out', required=True)
    return ap.parse_args()


def epoch_batches(ds, batch, epoch, seed, train):
    """Deterministic batches: seeded permutation per epoch (train, drop_last), fixed order (val)."""
    n = len(ds)
    order = np.random.default_rng(seed * 7919 + epoch).permutation(n) if train else np.arange(n)
    stop = n - n % batch if train else n
    for s in range(0, stop, batch):
        idx = order[s:s + batch]
        items = []
        for i in idx:
            ds.rng = np.random.default_rng((seed * 1_000_003 + epoch) * 1_000_003 + int(i))   # strong-view jitter
            items.append(ds[int(i)])
        yield {k: torch.stack([it[k] for it in items]) for k in items[0]}


def avg_dice(per_sample):
    """train_chaos.py:388-395: AvgMeter per class skips NaN; avg_all = mean of the class means, background excluded."""
    d = np.asarray(per_sample, dtype=np.float64)                       # (n, K)
    cls = [np.nanmean(d[:, c]) for c in range(1, d.shape[1])]
    return float(np.mean(cls)), [float(c) for c in cls]


def main():
    a = parse()
    torch.set_num_threads(a.threads)
    args = O.full_flags(epoch=a.epochs)
    sd = O.init_state(args, seed=a.seed)
    tr = SyntheticPhantoms(a.train, args.num_classes, size=a.size, do_strong=True, train=True, seed=a.seed)
    va = SyntheticPhantoms(a.val, args.num_classes, size=a.size, do_strong=False, train=False, seed=a.seed)
    snaps = {int(s) for s in a.snap.split(',') if s}

    if a.backend == 'hip':
        from pacingpseudo_amd import engine as E
        E.WINO_ENABLED, E.F16X3_ENABLED = bool(a.wino), bool(a.f16x3)
        from pacingpseudo_amd._lib import lib
        lib.pp_set_matrix_products(a.products)
        from pacingpseudo_amd.optim import FusedAdam
        from pacingpseudo_amd.utils import poly_lr_decay
        from tests.test_gpu_step import build_model
        args.storage = a.storage
        model = build_model(args, {k: v.numpy() for k, v in sd.items()})
        assert model.engine.storage == a.storage and model.engine.h16 == (a.storage != 'fp32')
        model.train()
        opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    else:
        adam = O.AdamState()
        if a.noise > 0:
            gen = torch.Generator().manual_seed(a.noise_seed)
            real = O.F.conv2d

            class _F:                                   # the oracle calls F.conv2d through its module attribute
                def __getattr__(self, k):
                    return getattr(torch.nn.functional, k)

                @staticmethod
                def conv2d(*x, **k):
                    y = real(*x, **k)
                    return y * (1 + a.noise * torch.randn(y.shape, generator=gen))
            O.F = _F()

    rows, step, bn_train, checks = [], 0, True, {}
    t_all = time.time()
    for ep in range(a.epochs):
        lr = O.lr_at(args.lr_decay, ep, args.epoch, args.lr)
        w = O.loss_weights(args, ep)
        t0 = time.time()
        last = float('nan')
        for b in epoch_batches(tr, a.batch, ep, a.seed, True):
            b = {k: v for k, v in b.items() if k not in ('label', 'label_strong')}
            if a.backend == 'hip':
                poly_lr_decay(opt, ep, args.epoch, args.lr)
                out = model({k: v.cuda() for k, v in b.items()}, mode='train', step=ep)
                loss = sum(out[k] * wt for k, wt in w.items())
                opt.zero_grad(); loss.backward(); opt.step()
                last = loss
            else:
                _, _, last = O.train_step(sd, b, ep, args, bn_train, adam, lr)
            step += 1
            if step in snaps:
                cur = ({k: v.detach().double().cpu() for k, v in model.state_dict().items()} if a.backend == 'hip'
                       else {k: v.double() for k, v in sd.items()})
                checks[step] = {k: [float(v.sum()), float(v.pow(2).sum())] for k, v in cur.items() if v.dtype.is_floating_point}
        if a.backend == 'hip':
            model.eval()
        bn_train = False                                   # train_chaos.py:370, never undone
        per = []
        for vb in epoch_batches(va, a.batch, 0, a.seed, False):
            with torch.no_grad():
                if a.backend == 'hip':
                    lg = model({k: v.cuda() for k, v in vb.items()}, mode='val')['segmentation/logits'].cpu()
                else:
                    lg = O.consistency_forward(sd, vb, 'val', None, args, training=False)['segmentation/logits']
            prob = torch.softmax(lg, 1).numpy()
            for i in range(len(prob)):
                per.append(O.compute_dice(prob[i], vb['label'][i].numpy()))
        d, cls = avg_dice(per)
        rows.append(dict(epoch=ep, dice=d, per_class=cls, loss=float(last), lr=lr, seconds=time.time() - t0))
        print(json.dumps(rows[-1]), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(dict(config=vars(a), rows=rows, checks=checks, done=False), open(a.out, 'w'), indent=1)
    json.dump(dict(config=vars(a), rows=rows, checks=checks, done=True, total_seconds=time.time() - t_all),
              open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
