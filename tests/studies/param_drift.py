"""Is the HIP-vs-CPU validation-Dice drift a bias or chaos?  Lock-step parameter-distance study (runs on the GPU box).

Eight trajectories start from the SAME weights and see the SAME batches (train_chaos.py semantics, BN eval from
epoch 1):  CPU reference (oracle, T1 threads) | CPU' (other thread count = other fp32 summation order) |
CPU + 1e-6 / 1e-5 relative noise on every conv output | HIP shipped kernels | HIP without Winograd | HIP without
split-fp16 | HIP pure fp32 direct (no Winograd, no split-fp16).  After every step the relative L2 distance of the
trainable parameters to the CPU reference is logged.  Reading: the same exponential growth
rate from a larger intercept = chaos amplifying rounding-level differences; a faster growth rate = a bias.

    python tests/studies/param_drift.py --steps 120 --out gpurun_out/param_drift.json"""
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
from tests.studies.dice_study import epoch_batches, avg_dice  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=64); ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--train', type=int, default=48); ap.add_argument('--val', type=int, default=16)
ap.add_argument('--steps', type=int, default=120); ap.add_argument('--epochs_sched', type=int, default=40)
ap.add_argument('--seed', type=int, default=1); ap.add_argument('--t1', type=int, default=16); ap.add_argument('--t2', type=int, default=5)
ap.add_argument('--out', default='gpurun_out/param_drift.json')
a = ap.parse_args()

args = O.full_flags(epoch=a.epochs_sched)
sd0 = O.init_state(args, seed=a.seed)
keys = O.trainable_keys(sd0)
stat_keys = [k for k in sd0 if k.endswith('running_mean') or k.endswith('running_var')]


class CpuTraj:
    def __init__(self, name, threads, noise=0.0):
        self.name, self.threads, self.noise = name, threads, noise
        self.sd = {k: v.clone() for k, v in sd0.items()}
        self.adam = O.AdamState()
        self.gen = torch.Generator().manual_seed(99)

    def step(self, b, ep, lr, bn_train):
        torch.set_num_threads(self.threads)
        realF = O.F
        if self.noise > 0:
            real, gen, noise = torch.nn.functional.conv2d, self.gen, self.noise

            class _F:
                def __getattr__(self, k):
                    return getattr(torch.nn.functional, k)

                @staticmethod
                def conv2d(*x, **k):
                    y = real(*x, **k)
                    return y * (1 + noise * torch.randn(y.shape, generator=gen))
            O.F = _F()
        try:
            O.train_step(self.sd, b, ep, args, bn_train, self.adam, lr)
        finally:
            O.F = realF

    def params(self):
        return self.sd

    def val_logits(self, vb):
        torch.set_num_threads(a.t1)
        return O.consistency_forward(self.sd, vb, 'val', None, args, training=False)['segmentation/logits']

    def set_eval(self):
        pass


class HipTraj:
    def __init__(self, name, wino, f16):
        from pacingpseudo_amd import engine as E
        from pacingpseudo_amd.optim import FusedAdam
        from tests.test_gpu_step import build_model
        self.name, self.flags = name, (wino, f16)
        self.E = E
        self.model = build_model(args, {k: v.numpy() for k, v in sd0.items()})
        self.model.train()
        self.opt = FusedAdam(self.model.parameters(), lr=args.lr, weight_decay=args.wd)

    def step(self, b, ep, lr, bn_train):
        self.E.WINO_ENABLED, self.E.F16X3_ENABLED = self.flags        # read when this model's plan is first built
        for g in self.opt.param_groups:
            g['lr'] = lr
        out = self.model({k: v.cuda() for k, v in b.items()}, mode='train', step=ep)
        w = O.loss_weights(args, ep)
        loss = sum(out[k] * wt for k, wt in w.items())
        self.opt.zero_grad(); loss.backward(); self.opt.step()

    def params(self):
        return {k: v.detach().cpu() for k, v in self.model.state_dict().items()}

    def val_logits(self, vb):
        self.E.WINO_ENABLED, self.E.F16X3_ENABLED = self.flags
        return self.model({k: v.cuda() for k, v in vb.items()}, mode='val')['segmentation/logits'].cpu()

    def set_eval(self):
        self.model.eval()


trajs = [CpuTraj('cpu_ref', a.t1), CpuTraj('cpu_other_threads', a.t2), CpuTraj('cpu_noise_1e-6', a.t1, 1e-6),
         CpuTraj('cpu_noise_1e-5', a.t1, 1e-5), HipTraj('hip_shipped', True, True), HipTraj('hip_no_wino', False, True),
         HipTraj('hip_no_f16x3', True, False), HipTraj('hip_fp32_direct', False, False)]
tr = SyntheticPhantoms(a.train, args.num_classes, size=a.size, do_strong=True, train=True, seed=a.seed)
va = SyntheticPhantoms(a.val, args.num_classes, size=a.size, do_strong=False, train=False, seed=a.seed)


def dist(p, ref, ks):
    num = sum(float((p[k].double() - ref[k].double()).pow(2).sum()) for k in ks)
    den = sum(float(ref[k].double().pow(2).sum()) for k in ks)
    return (num / den) ** 0.5


rows, step, ep, bn_train = [], 0, 0, True
t0 = time.time()
while step < a.steps:
    lr = O.lr_at(args.lr_decay, ep, args.epoch, args.lr)
    for b in epoch_batches(tr, a.batch, ep, a.seed, True):
        b = {k: v for k, v in b.items() if k not in ('label', 'label_strong')}
        for t in trajs:
            t.step({k: v.clone() for k, v in b.items()}, ep, lr, bn_train)
        step += 1
        ref = trajs[0].params()
        row = dict(step=step, epoch=ep)
        for t in trajs[1:]:
            p = t.params()
            row[t.name] = dist(p, ref, keys)
            row[t.name + ':bn_stats'] = dist(p, ref, stat_keys)
        rows.append(row)
        print(json.dumps(row), flush=True)
        if step >= a.steps:
            break
    for t in trajs:
        t.set_eval()
    bn_train = False
    dice = {}
    for t in trajs:
        per = []
        for vb in epoch_batches(va, a.batch, 0, a.seed, False):
            with torch.no_grad():
                prob = torch.softmax(t.val_logits(vb), 1).numpy()
            per += [O.compute_dice(prob[i], vb['label'][i].numpy()) for i in range(len(prob))]
        dice[t.name] = avg_dice(per)[0]
    rows.append(dict(step=step, epoch=ep, val_dice=dice))
    print(json.dumps(rows[-1]), flush=True)
    ep += 1
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(dict(config=vars(a), rows=rows, seconds=time.time() - t0), open(a.out, 'w'), indent=1)
