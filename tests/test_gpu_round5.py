"""Round 5: BatchNorm + LeakyReLU backward of the first layer with its weight gradient folded in (pp_bn_lrelu_bwd[_eval]_wgrad_c1).

Reference op: models/unet.py:188-193 on the one-channel input -- autograd of conv2d -> BatchNorm2d -> LeakyReLU wrt the
convolution weight and the BatchNorm parameters, in fp64 (torch), train mode (statistics per group) and eval mode."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _case(C, B, H, W, groups, seed):
    g = torch.Generator().manual_seed(seed)
    N = B * groups
    x = torch.randn(N, 1, H, W, generator=g, dtype=torch.float64)
    w = (torch.randn(C, 1, 3, 3, generator=g, dtype=torch.float64) / 3).requires_grad_(True)
    bias = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
    gamma[0] = -0.7
    gamma = gamma.requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_(True)
    rm = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    rv = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    dy = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    return x, w, bias, gamma, beta, rm, rv, dy


@pytest.mark.parametrize('storage', ['fp32', 'fp16'])
@pytest.mark.parametrize('C,B,H,W,groups,training', [(32, 2, 16, 16, 2, True), (32, 3, 32, 32, 1, True), (64, 1, 8, 12, 2, True),
                                                     (12, 2, 6, 10, 2, True), (32, 2, 16, 16, 2, False), (32, 64, 4, 4, 1, True)])
def test_first_layer_bn_backward_with_folded_weight_gradient(C, B, H, W, groups, training, storage):
    """dW, dgamma, dbeta (and the conv-bias gradient) of the fused call against fp64 autograd; train-mode statistics per group as
    two module calls would take them; ld_x = 4 (the packed one-channel image), a padded dy / z stride; tiny images (a block row
    spans several images); fp16 storage: the same through the _h16 entry point on fp16 copies of x, z and dy (compared with the
    autograd of those rounded tensors)."""
    from pacingpseudo_amd._lib import lib_for, stream_ptr
    h16 = storage == 'fp16'
    K = lib_for(2 if h16 else 4)
    adt = torch.float16 if h16 else torch.float32
    st = stream_ptr()
    dev = torch.device('cuda', 0)
    x, w, bias, gamma, beta, rm, rv, dy = _case(C, B, H, W, groups, C + H + groups)
    N = B * groups
    if h16:                         # the tensors the device will hold
        x = x.to(adt).double()
        dy = (dy * 1e-2).to(adt).double()
    z = F.conv2d(x, w, bias, 1, 1, 1)
    if h16:
        z = z.detach().to(adt).double()
        zin = z.clone().requires_grad_(True)
    else:
        zin = z
    ys = [F.leaky_relu(F.batch_norm(zin[gi * B:(gi + 1) * B], rm.clone(), rv.clone(), gamma, beta, training, 0.1, 1e-5), 0.01) for gi in range(groups)]
    torch.cat(ys).backward(dy)
    if h16:                         # z was rounded: chain the rest by hand
        dz_ref = zin.grad
        w_grad = torch.nn.grad.conv2d_weight(x, w.shape, dz_ref, 1, 1, 1)
        b_grad = dz_ref.sum((0, 2, 3))
    else:
        w_grad, b_grad = w.grad, bias.grad
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    ld = C + 4
    xd = torch.zeros(N, H, W, 4, device=dev, dtype=adt); xd[..., :1] = nhwc(x).to(dev).to(adt)
    xd[..., 1:] = 7.0                                           # the padding channels must not be read
    zd = torch.zeros(N, H, W, ld, device=dev, dtype=adt); zd[..., :C] = nhwc(z.detach()).to(dev).to(adt)
    dyd = torch.zeros(N, H, W, ld, device=dev, dtype=adt); dyd[..., :C] = nhwc(dy).to(dev).to(adt)
    coef = torch.empty(4, groups, C, device=dev)
    mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
    ppg = B * H * W
    nws = max(K.pp_bn_lrelu_bwd_wgrad_c1_workspace(C, ppg, groups), K.pp_bn_lrelu_bwd_wgrad_c1_workspace(C, N * H * W, 1))
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
    gd, bd = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    rmd, rvd = rm.float().to(dev), rv.float().to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    if training:
        K.pp_bn_train_stats(zd.data_ptr(), ld, C, ppg, groups, 1e-5, 0.1, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(),
                            nbt.data_ptr(), mean, invstd, scale, shift, ws.data_ptr(), nws, st)
    else:
        K.pp_bn_eval_coeffs(C, groups, 1e-5, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), mean, invstd, scale, shift, st)
    dw = torch.full((C, 1, 3, 3), 5.0, device=dev)
    dg, db, dbc = (torch.full((C,), 9.0, device=dev) for _ in range(3))
    tol = 3e-3 if h16 else TOL                                  # fp16: the LeakyReLU branch is read from a rounded z (a few flips)
    for acc in (0, 1):
        K.pp_bn_lrelu_bwd_wgrad_c1(dyd.data_ptr(), ld, zd.data_ptr(), ld, scale, shift, mean, invstd, gd.data_ptr(), 1 if training else 0,
                                   xd.data_ptr(), 4, H, W, dw.data_ptr(), acc, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, ppg, groups,
                                   0.01, ws.data_ptr(), nws, st)
        torch.cuda.synchronize()
        assert rel(dw, (acc + 1) * w_grad) < tol, (acc, rel(dw, (acc + 1) * w_grad))
    assert rel(dg, gamma.grad) < tol and rel(db, beta.grad) < tol
    if not training:
        assert rel(dbc, b_grad) < tol
        # the one-pass eval form on the stored output y
        yd = torch.zeros(N, H, W, ld, device=dev, dtype=adt)
        K.pp_bn_lrelu_fwd(zd.data_ptr(), ld, scale, shift, yd.data_ptr(), ld, C, ppg, groups, 0.01, st)
        dw2 = torch.full((C, 1, 3, 3), 5.0, device=dev)
        dg2, db2, dbc2 = (torch.full((C,), 9.0, device=dev) for _ in range(3))
        K.pp_bn_lrelu_bwd_eval_wgrad_c1(dyd.data_ptr(), ld, yd.data_ptr(), ld, scale, gd.data_ptr(), bd.data_ptr(), xd.data_ptr(), 4, H, W,
                                        dw2.data_ptr(), 0, dg2.data_ptr(), db2.data_ptr(), dbc2.data_ptr(), 0, C, N * H * W, 0.01,
                                        ws.data_ptr(), nws, st)
        torch.cuda.synchronize()
        tol2 = 2e-2 if h16 else TOL                             # (fp16: xhat is recovered from a rounded y)
        assert rel(dw2, w_grad) < tol2 and rel(db2, beta.grad) < tol2 and rel(dbc2, b_grad) < tol2 and rel(dg2, gamma.grad) < 5 * tol2


def test_first_layer_fold_in_the_training_step_is_the_separate_path():
    """The whole step with PP_FUSE_WG1=1 (default) against PP_FUSE_WG1=0 in fresh processes: same kernels everywhere else, and the
    folded weight gradient sums the same products in another order -- every parameter after three Adam steps within 1e-6 of the
    separate path (train mode, then eval mode), the first convolution's weight included."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, torch
from oracle import pacing_oracle as O
from tests.test_gpu_step import build_model
from pacingpseudo_amd.optim import FusedAdam
args = O.full_flags(init_ch=16, max_ch=64, hid_ch=16, feat_ch=[64, 64])
torch.manual_seed(1)
model = build_model(args)
opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=args.wd)
b = {k: v.cuda() for k, v in O.synthetic_batch(2, 64, 64, seed=5, keep=0.05).items() if k != 'label'}
model.train()
for i in range(4):
    if i == 2:
        model.eval()
    out = model(b, mode='train', step=0)
    loss = out['loss_pce'] + out['loss_ent'] * 0.1 + out['loss_cr'] * 0.1 + out['loss_aux_cls'] + out['loss_memory']
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
torch.save({'p': model.flat.params.cpu(), 'w0': model.backbone.state_dict()[next(k for k in model.backbone.state_dict() if k.endswith('weight'))].cpu()}, sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for v in ('1', '0'):
            out = os.path.join(td, f'p{v}.pt')
            env = dict(os.environ, PP_FUSE_WG1=v, PYTHONPATH=root)
            subprocess.run([sys.executable, '-c', code, out], check=True, env=env, cwd=root, timeout=600)
            res[v] = torch.load(out)
    a, b = res['1'], res['0']
    assert torch.isfinite(a['p']).all()
    assert not torch.equal(a['p'], torch.zeros_like(a['p']))
    d = float((a['p'] - b['p']).abs().max())
    assert d < 1e-5, d
    assert float((a['w0'] - b['w0']).abs().max()) < 1e-5


def test_weighted_loss_sum_equals_the_chain_of_torch_operations():
    """The one-launch loss assembly against train_chaos.py:273-310 written as torch operations: total and the five gradients
    bit for bit (fp32 products and sums in the same order), for awkward weights and magnitudes."""
    from pacingpseudo_amd.losses.losses import weighted_loss_sum
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(11)
    for trial in range(20):
        vals = (torch.randn(5, generator=g) * 10 ** float(torch.randint(-6, 4, (1,), generator=g))).tolist()
        ws = [1.0] + (torch.rand(4, generator=g) * 3).tolist()
        up = float(torch.randn((), generator=g))
        a = [torch.tensor(v, device=dev, dtype=torch.float32, requires_grad=True) for v in vals]
        b = [torch.tensor(v, device=dev, dtype=torch.float32, requires_grad=True) for v in vals]
        ref = a[0]
        for t, w in zip(a[1:], ws[1:]):
            ref = ref + t * w
        (ref * up).backward()
        got = weighted_loss_sum(b, ws)
        (got * up).backward()
        assert torch.equal(ref.detach(), got.detach()), (trial, float(ref), float(got))
        for x, y in zip(a, b):
            assert torch.equal(x.grad, y.grad)
    with pytest.raises(TypeError):
        weighted_loss_sum([torch.zeros((), device=dev, dtype=torch.float64)], [1.0])


def test_batched_weight_packs_equal_the_per_layer_calls():
    """pp_pack_conv3x3_weights_f16x3_batch / pp_wino_pack_weights_f16x3_batch (one launch for all layers) against the per-layer
    entry points: every packed buffer bit for bit -- ragged channel counts, a forward-only item (no data-gradient layout), more
    items than one launch carries (24)."""
    from pacingpseudo_amd._lib import PpPackItem, PpWinoPackItem, lib, stream_ptr
    st = stream_ptr()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(3)
    shapes = [(32, 1), (32, 32), (64, 32), (64, 64), (128, 64), (20, 12), (64, 192), (32, 96)] * 4          # (O, I): 32 items
    keep, items, ref = [], [], []
    for i, (O, I) in enumerate(shapes):
        ipad = (I + 3) // 4 * 4
        w = (torch.randn(O, I, 3, 3, generator=g) * 0.1).to(dev)
        with_b = (ipad == I and O % 4 == 0 and i % 5 != 0)
        wf, wb = torch.full((O, 9, ipad), 7.0, device=dev), (torch.full((I, 9, O), 7.0, device=dev) if with_b else None)
        wf2, wb2 = torch.full((O, 9, ipad), 7.0, device=dev), (torch.full((I, 9, O), 7.0, device=dev) if with_b else None)
        lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), O, I, ipad, wf2.data_ptr(), wb2.data_ptr() if with_b else None, st)
        items.append(PpPackItem(w.data_ptr(), O, I, ipad, wf.data_ptr(), wb.data_ptr() if with_b else None))
        keep.append(w); ref.append((wf, wb, wf2, wb2))
    arr = (PpPackItem * len(items))(*items)
    lib.pp_pack_conv3x3_weights_f16x3_batch(arr, len(items), st)
    torch.cuda.synchronize()
    for wf, wb, wf2, wb2 in ref:
        assert torch.equal(wf.view(torch.int32), wf2.view(torch.int32))
        assert wb is None or torch.equal(wb.view(torch.int32), wb2.view(torch.int32))
    wshapes = [(64, 256), (256, 256), (512, 256), (128, 384), (64, 1024), (72, 40)] * 5                       # 30 items
    items, ref = [], []
    for i, (O, I) in enumerate(wshapes):
        w = (torch.randn(O, I, 3, 3, generator=g) * 0.1).to(dev)
        with_b = i % 4 != 1
        uf, ub = torch.full((36, O, I), 7.0, device=dev), (torch.full((36, I, O), 7.0, device=dev) if with_b else None)
        uf2, ub2 = torch.full((36, O, I), 7.0, device=dev), (torch.full((36, I, O), 7.0, device=dev) if with_b else None)
        lib.pp_wino_pack_weights_f16x3(w.data_ptr(), O, I, 4, uf2.data_ptr(), ub2.data_ptr() if with_b else None, st)
        items.append(PpWinoPackItem(w.data_ptr(), O, I, uf.data_ptr(), ub.data_ptr() if with_b else None))
        keep.append(w); ref.append((uf, ub, uf2, ub2))
    arr = (PpWinoPackItem * len(items))(*items)
    lib.pp_wino_pack_weights_f16x3_batch(arr, len(items), st)
    torch.cuda.synchronize()
    for uf, ub, uf2, ub2 in ref:
        assert torch.equal(uf.view(torch.int32), uf2.view(torch.int32))
        assert ub is None or torch.equal(ub.view(torch.int32), ub2.view(torch.int32))


def test_batched_weight_packs_follow_a_rebuilt_parameter_slab():
    """The item tables of the batched packs hold raw weight pointers: when the flat parameter slab is rebuilt (``model.cuda()`` again,
    ``_flatten()``), the next step must pack from the NEW storage.  One optimizer step, then the gradients of a second step -- with
    the slab rebuilt in between (the old one filled with NaN) and without: bit-identical."""
    from oracle import pacing_oracle as O
    from tests.test_gpu_step import build_model
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags(init_ch=8, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    b = {k: v.cuda() for k, v in O.synthetic_batch(2, 64, 64, seed=7, keep=0.05).items() if k != 'label'}
    res = []
    for rebuild in (False, True):
        torch.manual_seed(1)
        model = build_model(args)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-2, weight_decay=0.0)
        out = model(b, mode='train', step=0)
        loss = out['loss_pce'] + out['loss_aux_cls']
        opt.zero_grad(); loss.backward(); opt.step()
        if rebuild:
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            old = model.flat.params
            model._flatten()                             # new slab, new weight addresses
            model.load_state_dict(sd)
            old.fill_(float('nan'))                      # a pack from the old storage would poison everything
            assert model.flat.params.data_ptr() != old.data_ptr()
        out = model(b, mode='train', step=0)
        loss = out['loss_pce'] + out['loss_aux_cls']
        for p_ in model.parameters():
            p_.grad = None
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), model.flat.grads.clone()))
    assert torch.isfinite(res[1][1]).all()
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


def test_training_driver_with_augmentation_on_its_own_stream_is_bit_identical(tmp_path, monkeypatch):
    """The driver uploads and augments a batch on a second stream (beside the previous training step; PP_AUG_STREAM, default 1)
    with non-blocking copies from the loader's pinned tensors and a pinned staging ring for the parameter tables.  Same kernels on the
    same data in the same order per stream: validation Dice per epoch and the final checkpoint are identical bit for bit to the run
    with everything on one stream (3 epochs, GPU augmentation with the strong view, a loader worker process per loader)."""
    import gc
    import glob
    import os
    import numpy as np
    from pacingpseudo_amd.train import train_main
    gc.collect()
    torch.cuda.empty_cache()          # the loader workers are forked from this process: keep what it has mapped small
    out = {}
    for tag, v in (('one', '0'), ('two', '1')):
        monkeypatch.setenv('PP_AUG_STREAM', v)
        root = str(tmp_path / tag)
        vd = train_main(['--tag', tag, '--session', 'Experiment', '--root', root, '--synthetic', '24', '--epoch', '3', '--batch_size', '4',
                         '--image_size', '64', '--num_workers', '1', '--do_loss_ent', '--do_decoder_consistency', '--do_aux_path',
                         '--do_memory', '--gpu_augment'])
        run = glob.glob(os.path.join(root, 't1', 'Experiment', f'Experiment-*-fold1-{tag}'))
        out[tag] = (vd, torch.load(os.path.join(run[0], 'ckps', 'ckp_2.pth'), map_location='cpu'))
    assert np.array_equal(out['one'][0], out['two'][0]), (out['one'][0], out['two'][0])
    for k, v in out['one'][1].items():
        assert torch.equal(v, out['two'][1][k]), k


def test_compact_evaluation_items_expand_to_the_one_hot_planes():
    """Evaluation items with uint8 class maps (`compact=True`: 6 bytes per pixel through the loader instead of 48) expanded on the
    device against the one-hot fp32 planes the reference's data set returns (chaos_dataset.py:92-105): image, label and scribble
    planes identical bit for bit, for two class counts, through collate_by_shape."""
    from pacingpseudo_amd.data import SyntheticPhantoms, collate_by_shape, expand_compact
    dev = torch.device('cuda', 0)
    for K, size in ((5, 64), (2, 48)):
        a = SyntheticPhantoms(6, K, size=size, train=False, native=True, seed=3)
        b = SyntheticPhantoms(6, K, size=size, train=False, native=True, compact=True, seed=3)
        ga = collate_by_shape([a[i] for i in range(6)])
        gb = collate_by_shape([b[i] for i in range(6)])
        assert len(ga) == len(gb)
        for x, y in zip(ga, gb):
            assert set(y) == {'image', 'label_idx', 'scribble_idx'} and y['label_idx'].dtype == torch.uint8
            e = expand_compact(y, K, dev)
            assert set(e) == {'image', 'label', 'scribble'}
            for k in ('image', 'label', 'scribble'):
                assert e[k].dtype == torch.float32 and torch.equal(e[k].cpu(), x[k]), k
        # a batch that is not compact passes through unchanged
        p = expand_compact(ga[0], K, dev)
        assert torch.equal(p['label'].cpu(), ga[0]['label'])
