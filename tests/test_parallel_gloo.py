"""Data-parallel plumbing on CPU (gloo, world_size 2): bucketed gradient all-reduce over the flat slab, global loss
denominators, memory-bank broadcast -- and the property that makes the sharding exact: with the denominators
all-reduced before the backward pass, the SUM of the per-rank gradients equals the single-process gradient of the
concatenated batch (BatchNorm in eval mode, the reference's state from epoch 1 on)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pacing_oracle as O
from tests import _golden as G


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny(args):
    from pacingpseudo_amd.models import ConsistencyRegulr
    return ConsistencyRegulr(
        kwargs_unet=dict(input_ch=1, init_ch=4, max_ch=32, num_classes=5, output_stride=8, is_stride_conv=False,
                         is_trans_conv=False, elab_end_points=True),
        kwargs_aux_path=dict(num_classes=5, feat_stage=args.feat_stage, feat_ch=args.feat_ch, hid_ch=8, aux_drop_prob=0.0,
                             do_memory=False, max_step=400, update_momentum=0.9, ensemble_mode='cosine_similarity'),
        args_parser=args)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from pacingpseudo_amd import parallel
    from pacingpseudo_amd.flat import FlatSlab
    w, r, _ = parallel.init_from_env('gloo')
    assert (w, r) == (world, rank)
    comm = parallel.Comm()
    args = O.default_args(init_ch=4, max_ch=32, hid_ch=8, feat_ch=[32, 32], do_loss_ent=True,
                          do_decoder_consistency=True, do_aux_path=True, do_memory=False)
    torch.manual_seed(1)
    model = _tiny(args)
    flat = FlatSlab([('backbone', [p for p in model.backbone.parameters() if p.requires_grad]),
                     ('aux_path', [p for p in model.aux_path.parameters() if p.requires_grad])])
    # ---- 1. buckets tile the slab exactly once, in the backward's completion order
    buckets = parallel.backbone_buckets(model)
    seen = [p for _, ps in buckets for p in ps]
    assert len(seen) == len(set(map(id, seen))) == len(flat.offsets)
    red = parallel.GradReducer(model, comm)
    covered = sorted(red._range(flat, tag) for tag, _ in buckets)
    a0, b0 = flat.segments['backbone']
    a1, b1 = flat.segments['aux_path']
    assert covered[0][0] == a0 and covered[-1][1] <= b1
    # ---- 2. semantic check: sum of per-rank gradients == gradient of the concatenated batch
    sd = O.init_state(args, seed=3)
    for k in sd:                                   # non-trivial BN statistics for the eval-mode forward
        if k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(len(k))) * 0.1
        if k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=torch.Generator().manual_seed(len(k))) + 0.5
    full = O.synthetic_batch(4, 32, 32, seed=5, keep=0.08)
    full['valid_mask'][0, :, :7] = 0               # unequal denominators on the two ranks
    local = {k: v[rank * 2:(rank + 1) * 2].clone() for k, v in full.items()}
    wts = O.loss_weights(args, 100)
    keys = O.trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    out = O.consistency_forward(sd, local, 'train', 100, args, training=False)
    n_lab = (local['scribble'].argmax(1) != args.ignored_index).sum().double()
    n_val = local['valid_mask'].sum().double()
    sums = torch.stack([n_lab, n_val])
    loc = sums.clone()
    comm.allreduce_sums(sums)                      # the denominators the engine all-reduces before finalising
    scale = {'loss_pce': loc[0] / sums[0], 'loss_ent': loc[1] / sums[1], 'loss_cr': loc[1] / sums[1],
             'loss_aux_cls': loc[0] / sums[0]}
    total = sum(out[name] * wt * scale[name].float() for name, wt in wts.items())
    total.backward()
    for name, p in model.named_parameters():
        if p in flat.grad_views:
            flat.grad_views[p].copy_(sd[name].grad)
    for tag, _ in buckets:
        red.bucket_ready(flat, tag)
    red.reduce(flat, ['backbone', 'aux_path'])
    if rank == 0:
        for k in keys:
            sd[k].grad = None
        out_f = O.consistency_forward(sd, full, 'train', 100, args, training=False)
        tot_f = sum(out_f[name] * wt for name, wt in wts.items())
        tot_f.backward()
        worst = 0.0
        for name, p in model.named_parameters():
            if p in flat.grad_views:
                ref = sd[name].grad
                if float(ref.abs().max()) < 1e-7:
                    continue
                worst = max(worst, G.rel_err(flat.grad_views[p].numpy(), ref.numpy()))
        q.put(('grad_err', worst))
    # ---- 2b. BatchNorm buffers after an epoch of per-rank statistics: averaged / aligned before the eval switch
    bnm = torch.nn.Sequential(torch.nn.Conv2d(1, 3, 1), torch.nn.BatchNorm2d(3))
    with torch.no_grad():
        bnm[1].running_mean.fill_(float(rank + 1))
        bnm[1].running_var.fill_(float(2 * rank + 1))
        bnm[1].num_batches_tracked.fill_(7 + rank)
    parallel.sync_bn_buffers(bnm)
    assert torch.allclose(bnm[1].running_mean, torch.full((3,), 1.5)) and torch.allclose(bnm[1].running_var, torch.full((3,), 2.0))
    assert int(bnm[1].num_batches_tracked) == 7
    # ---- 3. memory bank broadcast from rank 0
    bank = torch.full((5, 8, 1, 1), float(rank + 1))
    comm.broadcast_bank(torch.nn.Parameter(bank, requires_grad=False))
    assert float(bank.mean()) == 1.0
    # ---- 4. sharded validation (train.py): every rank scores its share, ONE all-reduce of the device-side meters; the result must
    # be the whole set's per-class means of the non-NaN samples and the n-weighted loss (train_chaos.py:383-395)
    from pacingpseudo_amd.utils.metrics import ValAccumulator
    va = ValAccumulator(3, 'cpu')
    K = 3
    # rank 0: two samples, class 2 absent in both (NaN -> not counted); rank 1: three samples
    dice_sum = [torch.tensor([1.5, 0.9, 0.0]), torch.tensor([2.4, 1.2, 0.7])][rank]
    count = [torch.tensor([2.0, 2.0, 0.0]), torch.tensor([3.0, 3.0, 1.0])][rank]
    va.acc[:K] += dice_sum.double()
    va.acc[K:2 * K] += count.double()
    va.acc[2 * K] += [0.8 * 2, 0.5 * 3][rank]
    va.acc[2 * K + 1] += [2, 3][rank]
    avg, loss, n = va.result(parallel.all_reduce_sum)
    assert n == 5 and abs(loss - (0.8 * 2 + 0.5 * 3) / 5) < 1e-12
    assert np.allclose(avg, [3.9 / 5, 2.1 / 5, 0.7 / 1], atol=1e-12)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(540)
        assert p.exitcode == 0, f'rank process failed with exit code {p.exitcode}'
    name, err = q.get(timeout=5)
    assert name == 'grad_err' and err < 2e-4, err


def test_gradient_buckets_cover_the_strided_transposed_variant():
    """The transposed-convolution weights of --is_trans_conv are parameters of the decoder blocks: the data-parallel gradient
    buckets must tile the flat slab of that variant exactly once too (no process group needed for this check)."""
    from pacingpseudo_amd import parallel
    from pacingpseudo_amd.flat import FlatSlab
    from pacingpseudo_amd.models import ConsistencyRegulr
    args = O.default_args(init_ch=4, max_ch=32, hid_ch=8, feat_ch=[32, 32], do_loss_ent=True, do_decoder_consistency=True,
                          do_aux_path=True, do_memory=False)
    model = ConsistencyRegulr(
        kwargs_unet=dict(input_ch=1, init_ch=4, max_ch=32, num_classes=5, output_stride=8, is_stride_conv=True, is_trans_conv=True,
                         elab_end_points=True),
        kwargs_aux_path=dict(num_classes=5, feat_stage=args.feat_stage, feat_ch=args.feat_ch, hid_ch=8, aux_drop_prob=0.0,
                             do_memory=False, max_step=400, update_momentum=0.9, ensemble_mode='cosine_similarity'),
        args_parser=args)
    flat = FlatSlab([('backbone', [p for p in model.backbone.parameters() if p.requires_grad]),
                     ('aux_path', [p for p in model.aux_path.parameters() if p.requires_grad])])
    seen = [p for _, ps in parallel.backbone_buckets(model) for p in ps]
    assert len(seen) == len(set(map(id, seen))) == len(flat.offsets)
    assert any(p is model.backbone.dec_block3.up_samp.weight for p in seen)
