"""The whole training iteration replayed from a hipGraph (pacingpseudo_amd/graph.py) against the eager iteration.

The iteration body is train_chaos.py:263-315; GraphedStep captures forward, loss assembly, backward (two streams) and the fused
optimizer once and replays them.  Same kernels, same buffers, same order: everything must agree with the eager run BIT FOR BIT.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests.test_gpu_step import build_model  # noqa: E402


def _loss_fn(args):
    from pacingpseudo_amd.utils import gaussian_ramp_up

    def f(out, epoch):
        loss = out['loss_pce']
        loss = loss + out['loss_ent'] * gaussian_ramp_up(epoch, args.loss_ent_weight, scale=args.ramp_up_scale)
        loss = loss + out['loss_cr'] * gaussian_ramp_up(epoch, args.loss_cr_weight, scale=args.ramp_up_scale)
        loss = loss + out['loss_aux_cls'] * args.loss_aux_weight + out['loss_memory'] * args.loss_memory_weight
        return loss
    return f


def _eager_step(model, opt, f, batch, epoch):
    out = model(batch, mode='train', step=epoch)
    loss = f(out, epoch)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach().clone()


@pytest.mark.parametrize('storage', ['fp32', 'fp16'])
def test_graph_replay_equals_eager_steps(storage):
    """Seven iterations (epochs 0,0,0,0,1,1,1: the epoch change re-captures -- ramp-up weights and the memory-bank momentum are
    baked into a capture; BatchNorm switches to eval mode at the epoch change as in train_chaos.py:370; the learning rate changes
    on the host between steps as poly_lr_decay does) -- eager against GraphedStep: losses, parameters, Adam state and step counts,
    BatchNorm buffers and the memory bank identical bit for bit.  Full widths (the split-fp16 kernels and, for 'fp16', the
    16-bit storage plan with its loss scale and overflow guard), 128x128, both streams."""
    from pacingpseudo_amd.graph import GraphedStep
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags()
    args.storage = storage
    f = _loss_fn(args)
    batches = [{k: v.cuda() for k, v in O.synthetic_batch(2, 128, 128, seed=s, keep=0.05).items() if k != 'label'} for s in (3, 4)]
    epochs = [0, 0, 0, 0, 1, 1, 1]
    lrs = [1e-3, 1e-3, 5e-4, 5e-4, 2e-4, 2e-4, 1e-4]
    runs = {}
    for tag in ('eager', 'graph'):
        torch.manual_seed(1)
        model = build_model(args)
        opt = FusedAdam(model.parameters(), lr=lrs[0], weight_decay=args.wd)
        gs = GraphedStep(model, opt, f, warmup=1) if tag == 'graph' else None
        model.train()
        losses = []
        for i, ep in enumerate(epochs):
            if ep == 1 and epochs[i - 1] == 0:
                model.eval()                                   # train_chaos.py:370, never undone
            for g in opt.param_groups:
                g['lr'] = lrs[i]
            b = batches[i % 2]
            if gs is None:
                losses.append(_eager_step(model, opt, f, b, ep))
            else:
                loss, _ = gs(b, ep)
                losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        sd = opt.state_dict()['slabs'][0]
        runs[tag] = dict(losses=torch.stack(losses).cpu(), params=model.flat.params.clone(), m=sd['m'], v=sd['v'], steps=sd['steps'],
                         state={k: v.detach().clone() for k, v in model.state_dict().items()})
        if gs is not None:
            assert gs.captures == 2 and gs.replays == len(epochs) - 1, (gs.captures, gs.replays)
    e, g = runs['eager'], runs['graph']
    assert torch.isfinite(e['losses']).all()
    assert torch.equal(e['losses'], g['losses']), (e['losses'], g['losses'])
    assert torch.equal(e['params'], g['params'])
    assert torch.equal(e['m'], g['m']) and torch.equal(e['v'], g['v']) and e['steps'] == g['steps'] == {'backbone': 7, 'aux_path': 7}
    for k, v in e['state'].items():
        assert torch.equal(v, g['state'][k]), k


def test_inference_after_graph_replay_sees_the_new_weights():
    """A replay updates the weights without any host-side optimizer call: the forward-only plans must still notice
    (GraphedStep bumps the slab version the packed-weight cache watches)."""
    from pacingpseudo_amd.graph import GraphedStep
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags(init_ch=8, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    torch.manual_seed(1)
    model = build_model(args)
    opt = FusedAdam(model.parameters(), lr=1e-2, weight_decay=0.0)
    gs = GraphedStep(model, opt, _loss_fn(args), warmup=1)
    b = {k: v.cuda() for k, v in O.synthetic_batch(2, 64, 64, seed=5, keep=0.05).items() if k != 'label'}
    model.train()
    gs(b, 0)
    gs(b, 0)                                               # capture + first replay

    def val():
        with torch.no_grad():
            return model(b, mode='val')['segmentation/logits'].clone()
    a0 = val()
    gs(b, 0)                                               # a replay: weights change
    a1 = val()
    assert not torch.equal(a0, a1)
    # reference: a fresh model with the same state must give the same validation logits
    torch.manual_seed(1)
    ref = build_model(args)
    ref.load_state_dict(model.state_dict())
    ref.train(model.training)
    with torch.no_grad():
        r = ref(b, mode='val')['segmentation/logits']
    assert torch.equal(a1, r)


def test_training_driver_with_graph_step_equals_the_eager_driver(tmp_path):
    """train_chaos.py --graph_step (the iteration replayed from a hipGraph captured once per epoch) against the same run without it:
    the validation Dice per epoch and the final checkpoint must be identical bit for bit (3 epochs: train-mode BN in epoch 0, eval
    mode afterwards, the poly learning rate changing every epoch)."""
    import glob
    import os
    import numpy as np
    from pacingpseudo_amd.train import train_main
    out = {}
    for tag, extra in (('eager', []), ('graph', ['--graph_step'])):
        root = str(tmp_path / tag)
        vd = train_main(['--tag', tag, '--session', 'Experiment', '--root', root, '--synthetic', '16', '--epoch', '3', '--batch_size', '4',
                         '--image_size', '64', '--num_workers', '0', '--cpu_input', '--do_loss_ent', '--do_decoder_consistency',
                         '--do_aux_path', '--do_memory'] + extra)
        run = glob.glob(os.path.join(root, 't1', 'Experiment', f'Experiment-*-fold1-{tag}'))
        sd = torch.load(os.path.join(run[0], 'ckps', 'ckp_2.pth'), map_location='cpu')
        out[tag] = (vd, sd)
    assert np.array_equal(out['eager'][0], out['graph'][0]), (out['eager'][0], out['graph'][0])
    for k, v in out['eager'][1].items():
        assert torch.equal(v, out['graph'][1][k]), k
