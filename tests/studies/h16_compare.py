"""16-bit storage mode against the fp32 path on the same weights and batch (GPU): loss / logits / gradient differences and
step time.   python tests/studies/h16_compare.py [--batch 4] [--size 256] [--steps 10] [--scale 1024]
Writes gpurun_out/h16_compare.json."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--eval-bn', action='store_true')
    cli = ap.parse_args()
    import torch
    from bench import build, train_iteration
    from pacingpseudo_amd.data import full_flags, synthetic_batch
    from pacingpseudo_amd.optim import FusedAdam
    dev = torch.device('cuda', 0)
    a32 = full_flags()
    a16 = full_flags()
    a16.storage = 'fp16'
    m32 = build(a32, dev)
    m16 = build(a16, dev)
    m16.load_state_dict(m32.state_dict())
    assert m16.engine.h16 and not m32.engine.h16
    batch = {k: v.to(dev) for k, v in synthetic_batch(cli.batch, cli.size, cli.size, a32.num_classes, seed=0).items() if k != 'label'}
    res = {}
    outs = {}
    for name, m, a in (('fp32', m32, a32), ('h16', m16, a16)):
        m.train()
        if cli.eval_bn:
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.eval()
        torch.manual_seed(0)
        out = m(batch, mode='train', step=0)
        loss = out['loss_pce'] + out['loss_ent'] + out['loss_cr'] + out['loss_aux_cls'] + out['loss_memory']
        for p in m.parameters():
            p.grad = None
        loss.backward()
        torch.cuda.synchronize()
        outs[name] = dict(out={k: v.detach().float().clone() for k, v in out.items()},
                          grads={n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    o32, o16 = outs['fp32'], outs['h16']
    for k in o32['out']:
        d = (o32['out'][k] - o16['out'][k]).abs().max().item()
        r = o32['out'][k].abs().max().item()
        res[f'out/{k}'] = dict(max_abs_diff=d, ref_max=r)
    worst = []
    for n, g in o32['grads'].items():
        h = o16['grads'][n]
        rel = ((g - h).norm() / (g.norm() + 1e-30)).item()
        cos = (torch.dot(g.flatten(), h.flatten()) / (g.norm() * h.norm() + 1e-30)).item()
        worst.append((rel, cos, n, g.norm().item()))
    worst.sort(reverse=True)
    res['grads_worst'] = [dict(name=n, rel_l2=r, cos=c, norm=gn) for r, c, n, gn in worst[:8]]
    res['grads_median_rel_l2'] = worst[len(worst) // 2][0]
    res['grads_all'] = {n: round(o, 5) for n, o in ((n, ((g - o16['grads'][n]).norm() / (g.norm() + 1e-30)).item()) for n, g in o32['grads'].items()) if n.endswith('conv.weight') or 'fc' in n or 'final' in n}
    res['nonfinite'] = bool(any(not torch.isfinite(g).all() for g in o16['grads'].values()))
    # step time
    for name, m, a in (('fp32', m32, a32), ('h16', m16, a16)):
        opt = FusedAdam(m.parameters(), lr=a.lr, weight_decay=a.wd)
        for _ in range(3):
            train_iteration(m, opt, batch, a, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(cli.steps):
            train_iteration(m, opt, batch, a, 0)
        torch.cuda.synchronize()
        res[f'ms_per_step/{name}'] = (time.perf_counter() - t0) / cli.steps * 1e3
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/h16_compare.json', 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
