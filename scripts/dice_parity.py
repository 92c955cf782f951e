"""Validation-Dice parity after equal steps: the HIP path on the GPU vs the CPU oracle (= reference path), same
initial weights, same synthetic-phantom data, same schedule (train_chaos.py semantics incl. the BN eval switch).
A SECOND CPU trajectory with a different thread count (= different fp32 summation order inside oneDNN) is trained
alongside: it is the yardstick for how far two runs of the reference path itself drift apart on this tiny,
chaotic problem.  Prints one line per epoch and a JSON summary (kept under profiles/)."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pacing_oracle as O
from pacingpseudo_amd.data import SyntheticPhantoms
from pacingpseudo_amd.optim import FusedAdam
from pacingpseudo_amd.utils import poly_lr_decay
from pacingpseudo_amd.utils.metrics import batch_dice
from tests.test_gpu_step import build_model

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=64); ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--train', type=int, default=48); ap.add_argument('--val', type=int, default=16)
ap.add_argument('--epochs', type=int, default=6); ap.add_argument('--total_epochs', type=int, default=400)
ap.add_argument('--out', default='gpurun_out/dice_parity.json')
a = ap.parse_args()
torch.set_num_threads(min(32, os.cpu_count() or 1))
args = O.full_flags(epoch=a.total_epochs)
torch.manual_seed(1)
model = build_model(args)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
sd2 = {k: v.clone() for k, v in sd.items()}
adam2 = O.AdamState()
T1, T2 = min(32, os.cpu_count() or 1), 5
opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
adam = O.AdamState()

def loader(train):
    ds = SyntheticPhantoms(a.train if train else a.val, 5, size=a.size, do_strong=train, train=train, seed=1)
    return torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=False, num_workers=0, drop_last=train)

def dice_of(logits, label):
    d = batch_dice(logits.cuda(), label.cuda())
    per = [np.nanmean(d[:, c]) for c in range(1, 5)]
    return float(np.mean(per))

rows = []
bn_train = True
for ep in range(a.epochs):
    opt, lr = poly_lr_decay(opt, ep, args.epoch, args.lr)
    w = O.loss_weights(args, ep)
    t0 = time.time()
    for gb, cb in zip(loader(True), loader(True)):
        gb.pop('label'); gb.pop('label_strong', None); cb.pop('label_strong', None)
        out = model({k: v.cuda() for k, v in gb.items()}, mode='train', step=ep)
        loss = sum(out[k] * wt for k, wt in w.items())
        opt.zero_grad(); loss.backward(); opt.step()
        cbb = {k: v for k, v in cb.items() if k != 'label'}
        torch.set_num_threads(T1)
        O.train_step(sd, cbb, ep, args, bn_train, adam, lr)
        torch.set_num_threads(T2)
        O.train_step(sd2, {k: v.clone() for k, v in cbb.items()}, ep, args, bn_train, adam2, lr)
        torch.set_num_threads(T1)
    model.eval(); bn_train = False                       # train_chaos.py:370, never undone
    dg, dc, dc2, n = 0.0, 0.0, 0.0, 0
    for vb in loader(False):
        with torch.no_grad():
            lg = model({k: v.cuda() for k, v in vb.items()}, mode='val')['segmentation/logits']
            lc = O.consistency_forward(sd, vb, 'val', None, args, training=False)['segmentation/logits']
            lc2 = O.consistency_forward(sd2, vb, 'val', None, args, training=False)['segmentation/logits']
        dg += dice_of(lg, vb['label']) * len(lg); dc += dice_of(lc, vb['label']) * len(lg); n += len(lg)
        dc2 += dice_of(lc2, vb['label']) * len(lg)
        agree = float((lg.argmax(1).cpu() == lc.argmax(1)).float().mean())
    rows.append(dict(epoch=ep, dice_hip=dg / n, dice_cpu=dc / n, dice_cpu_other_threads=dc2 / n,
                     diff_pt=100 * (dg - dc) / n, cpu_vs_cpu_diff_pt=100 * (dc2 - dc) / n, argmax_agreement=agree,
                     loss_hip=float(loss), seconds=time.time() - t0))
    print(rows[-1], flush=True)
res = dict(config=vars(a), cpu_threads=[T1, T2], rows=rows, max_abs_diff_pt=max(abs(r['diff_pt']) for r in rows),
           max_abs_cpu_vs_cpu_diff_pt=max(abs(r['cpu_vs_cpu_diff_pt']) for r in rows),
           mean_dice_last_half=dict(hip=float(np.mean([r['dice_hip'] for r in rows[len(rows) // 2:]])),
                                    cpu=float(np.mean([r['dice_cpu'] for r in rows[len(rows) // 2:]])),
                                    cpu_other=float(np.mean([r['dice_cpu_other_threads'] for r in rows[len(rows) // 2:]]))))
json.dump(res, open(a.out, 'w'), indent=1)
print('max |Dice_hip - Dice_cpu| =', res['max_abs_diff_pt'], 'points;  max |Dice_cpu(T2) - Dice_cpu(T1)| =',
      res['max_abs_cpu_vs_cpu_diff_pt'], 'points;  mean Dice over the last half:', res['mean_dice_last_half'])
