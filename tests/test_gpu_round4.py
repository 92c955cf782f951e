"""Round-4 GPU parity cases.

* gradient parity at the BENCHMARK's launch geometry: 32 images per view (64 images per conv / weight-gradient launch),
  256x256, full flags, one train-mode-BN step -- every parameter gradient against the branch-aligned oracle.  The
  weight-gradient kernels choose their reduction splits from the batch (pp_wino.hip wino_wg_plan, pp_conv.hip
  wgrad_h16_blocks), so the 2-image test of round 2 did not exercise the configuration bench.py times.
* the plan cache keeps the training plan across validation shapes.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests.test_gpu_step import (TOL_GRAD, TOL_OUT, build_model, check_grads, iteration,  # noqa: E402
                                 oracle_with_device_branches)



@pytest.mark.timeout(2400)
@pytest.mark.parametrize('bn_training', [True, False])
def test_benchmark_batch_gradients_against_oracle(bn_training):
    """bench.py's exact configuration (BASELINE.json configs[1]): batch 32 per view at 256x256, full flags, BN in train
    mode.  Outputs 1e-4 against the fp32 oracle, element-wise violation share of the logits <= 1e-4; every parameter gradient
    1e-4 against the oracle evaluated in fp64 with the device's LeakyReLU / max-pool choices, and 2e-4 (+ the fp32 oracle's
    own summation noise, measured in the test) against the fp32 oracle.
    bn_training = False (round 5): the same with BatchNorm in EVAL mode -- the reference's state for 399 of
    its 400 epochs (train_chaos.py:370: model.eval() after epoch 0, never undone), where the convolution epilogues normalise and
    activate and the backward is the one-pass form; running statistics made non-trivial by three train-mode forwards first."""
    from pacingpseudo_amd._lib import lib
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags()
    torch.manual_seed(1)
    model = build_model(args)
    batch = O.synthetic_batch(32, 256, 256, seed=0)          # bench.py's batch (pacingpseudo_amd/data.py:synthetic_batch)
    epoch = 0
    if not bn_training:
        bdev = {k: v.cuda() for k, v in batch.items() if k != 'label'}
        with torch.no_grad():
            for _ in range(3):
                model(bdev, mode='train', step=0)          # (train-mode BN under no_grad still updates the running statistics)
        del bdev
        model.eval()
        epoch = 1
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
    sd_start = {k: v.clone() for k, v in sd.items()}
    rec, grads = iteration(model, opt, batch, args, epoch)
    plan = model.engine.last_plan
    # the plan is the benchmark's: 64 images per launch, Winograd F(4x4) split-fp16 GEMMs on the 32x32 maps, split-fp16
    # halo kernels above; the weight-gradient split counts quoted here are functions of exactly this shape
    assert (plan.Bt, plan.H, plan.W, plan.G) == (64, 256, 256, 2)
    assert plan.wino['dec_block5.conv_block.conv_layer1'] and plan.wino16_wg['enc_block6.conv_block.conv_layer1']
    assert plan.f16['dec_block1.conv_block.conv_layer1'] and plan.f16['enc_block2.conv_block.conv_layer2']
    splits = {n: lib.pp_conv3x3_wino_bwd_weight_splits(L.cout, L.cin, 64, 32, 32, L.dil)
              for n, L in ((L.name, L) for L in model.engine.layers) if plan.wino[n] and plan.sizes[3] == (32, 32)
              and L.name != 'dec_block3.conv_block.conv_layer1'}
    splits2 = {n: lib.pp_conv3x3_wino_bwd_weight_splits(L.cout, L.cin, 4, 32, 32, L.dil)
               for n, L in ((L.name, L) for L in model.engine.layers) if n in splits}
    print('Winograd weight-gradient reduction splits at 64 images per launch:', splits, '; at 4 images:', splits2)
    assert splits != splits2, 'the batch-32 plan must differ from the 2-image plan of test_benchmark_shape_step_against_oracle'

    mode = 'train-mode BN' if bn_training else 'eval-mode BN'
    ref_out, ref_grads, ref_total = O.train_step(sd, batch, epoch, args, training=bn_training)
    for k, v in ref_out.items():
        if k.startswith('_') or not torch.is_tensor(v):
            continue
        e = G.rel_err(rec[k].double().cpu().numpy(), v.numpy())
        assert e < TOL_OUT, f'{k}: rel err {e:.3e}'
    assert abs(float(rec['total_loss']) - ref_total) < TOL_OUT * max(1.0, abs(ref_total))
    for key in ('segmentation/logits', 'segmentation/logits_strong'):
        G.argmax_report(rec[key].cpu().numpy(), ref_out[key].numpy(), f'batch 32 at 256x256 step, {mode}, {key}')
        r = G.elementwise_report(rec[key].double().cpu().numpy(), ref_out[key].numpy(), f'batch 32 at 256x256 step, {mode}, {key}')
        assert r['violation_share'] < G.TOL_VIOLATION_SHARE, r
    del ref_out
    # Gradients.  At 64 images x 65,536 pixels the fp32 CPU path itself is no longer a 2e-4 reference: its weight gradients of
    # the 256x256 layers and the head bias are sums of 4.2 M largely cancelling fp32 terms (tests/studies/grad_noise_b32.py on
    # the GPU box, r04: oracle fp32 vs oracle fp64 2.2e-4 on enc_block1.conv_layer2.conv.weight, 3.0e-4 on final_conv.bias;
    # HIP vs oracle fp64 <= 1.9e-5 on EVERY parameter).  The gate therefore compares with the oracle evaluated in fp64 (same
    # starting state, same inputs, the device's LeakyReLU / max-pool choices) at 1e-4, and the fp32 oracle gets the bound
    # its own noise allows.
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd_start.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    _, og64, _ = oracle_with_device_branches(model, sd64, b64, epoch, args, bn_training)
    og_np = {k: v.numpy() for k, v in og64.items() if v is not None}
    worst = check_grads(grads, og_np, bn_training, tag=f'batch 32 vs fp64 oracle, {mode} ', tol=1e-4)
    # The fp32 oracle with the device's choices: the bound its own summation noise allows.  One BN mode only (round 6, VERDICT r05
    # item 4: each oracle evaluation at this size is 20 - 50 s of host time and the suite has a wall-clock limit) -- the fp64 gate
    # above is the tight one in both modes, and the eval-mode step is compared with the fp32 oracle raw below.
    if bn_training:
        _, og32, _ = oracle_with_device_branches(model, sd_start, batch, epoch, args, bn_training)
        # (eval-mode BN: the convolution biases in front of BatchNorm have real gradients -- sums over 4.2 M pixels like the others)
        noise = {k: G.rel_err(og32[k].double().numpy(), og_np[k]) for k in og_np if not (bn_training and G.is_bias_before_bn(k))}
        worst32 = check_grads(grads, {k: v.numpy() for k, v in og32.items() if v is not None}, bn_training, tag=f'batch 32 vs fp32 oracle, {mode} ',
                              tol=TOL_GRAD, tols={k: TOL_GRAD + 2.0 * n for k, n in noise.items()})
        del og32
    else:
        noise, worst32 = {'-': float('nan')}, (float('nan'), 'not evaluated in eval mode (see the train-mode case)')
    G._report(dict(kind='gradients', tag=f'batch 32 at 256x256, {mode}, aligned', vs_fp64_oracle=worst[0], vs_fp64_key=worst[1],
                   vs_fp32_oracle=worst32[0], vs_fp32_key=worst32[1], fp32_oracle_own_noise=max(noise.values()),
                   tol_fp64=1e-4, wino_wgrad_splits=splits))
    # raw (unaligned) gradients: with 64 images the sums are no longer dominated by single activations on the kink
    raw = check_grads(grads, {k: v.numpy() for k, v in ref_grads.items() if v is not None}, bn_training,
                      tag=f'batch 32 raw, {mode} ', tol=G.TOL_GRAD_RAW_LARGE_BATCH)
    G._report(dict(kind='gradients', tag=f'batch 32 at 256x256, {mode}, raw', worst_rel_err=raw[0], worst_key=raw[1]))


def test_lazy_batchnorm_forms_match_the_separate_pass():
    """Train-mode BatchNorm + LeakyReLU applied by the CONSUMER while it loads (pp_*_lazy: the 1x1 head, the two-half halo kernel
    and the halo-tile weight gradients -- the forms that measured faster; the Winograd / max-pool / bilinear forms of round 4 did
    not and were removed in round 5) against the separate normalise + activate pass: one full-flags step of the full-width
    network at 128x128, the lazy forms on (default), the halo form off, all off."""
    from pacingpseudo_amd import engine as E
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags()
    batch = O.synthetic_batch(2, 128, 128, seed=7, keep=0.05)
    saved = (E.LAZY_BN, E.LAZY_HALO)
    runs = {}
    try:
        for tag, flags in (('off', (False, False)), ('all', (True, False)), ('default', saved)):      # 'all': head only
            E.LAZY_BN, E.LAZY_HALO = flags
            torch.manual_seed(1)
            model = build_model(args)
            opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
            rec, grads = iteration(model, opt, batch, args, 0)
            plan = model.engine.last_plan
            runs[tag] = (rec, grads, {k: v for k, v in plan.lazy_out.items() if v},
                         {k: v.detach().clone() for k, v in model.state_dict().items() if 'running' in k})
    finally:
        E.LAZY_BN, E.LAZY_HALO = saved
    assert not runs['off'][2]
    assert set(runs['all'][2]) == {'dec_block1.conv_block.conv_layer2'}, sorted(runs['all'][2])
    assert 'dec_block1.conv_block.conv_layer2' in runs['default'][2]          # the 1x1 head reads its input lazily by default
    # ... and so do the second convolutions of the narrow DoubleConvs (two-half halo kernel + halo-tile weight gradient)
    assert {'enc_block1.conv_block.conv_layer1', 'enc_block2.conv_block.conv_layer1', 'dec_block2.conv_block.conv_layer1',
            'dec_block1.conv_block.conv_layer1'} <= set(runs['default'][2]), sorted(runs['default'][2])
    ref_rec, ref_grads = runs['off'][0], runs['off'][1]
    for tag in ('all', 'default'):
        rec, grads, _, stats = runs[tag]
        for k, v in ref_rec.items():
            assert G.rel_err(rec[k].double().cpu().numpy(), v.double().cpu().numpy()) < 2e-6, (tag, k)
        errs = sorted(((G.rel_err(grads[k].double().cpu().numpy(), v.double().cpu().numpy()), k) for k, v in ref_grads.items()
                       if v is not None and not G.is_bias_before_bn(k)), reverse=True)
        # same arithmetic element for element (one fma + select, pp_lazy_apply4 == bn_lrelu_fwd_kernel); what differs is
        # the order of fp32 sums downstream of a different kernel selection: nothing
        assert errs[0][0] < 5e-6, (tag, [(f'{e:.2e}', k) for e, k in errs[:6]])
        for k, v in runs['off'][3].items():
            assert torch.equal(stats[k], v), (tag, k)


def test_plan_cache_keeps_training_plan_across_validation_shapes():
    """validation at native slice sizes creates one plan per shape; the cache must be LRU and never evict the
    plan of the live training step (train -> five validation shapes -> train again: no rebuild)."""
    args = O.full_flags(init_ch=8, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    torch.manual_seed(2)
    model = build_model(args)
    eng = model.engine
    tb = {k: v.cuda() for k, v in O.synthetic_batch(2, 64, 64, seed=1, keep=0.05).items() if k != 'label'}
    out = model(tb, mode='train', step=0)
    (out['loss_pce'] + out['loss_cr']).backward()
    train_plan = eng.last_plan
    built = eng.plans_built
    model.eval()
    shapes = [(1, 32, 32), (2, 40, 24), (1, 48, 64), (3, 32, 64), (1, 64, 32), (2, 24, 24), (1, 32, 32)]
    with torch.no_grad():
        for (b, h, w) in shapes:
            vb = {k: v.cuda() for k, v in O.synthetic_batch(b, h, w, seed=2, keep=0.05).items() if k != 'label'}
            model(vb, mode='val')
            assert not eng.last_plan.trainable, 'no-grad calls must run on light plans (no gradient buffers)'
    assert eng.plans_built == built + 6, (eng.plans_built, built)        # the repeated (1, 32, 32) is a cache hit
    model.train()
    out = model(tb, mode='train', step=0)
    (out['loss_pce'] + out['loss_cr']).backward()
    assert eng.last_plan is train_plan and eng.plans_built == built + 6, 'the training plan was rebuilt'
    torch.cuda.synchronize()


@pytest.mark.parametrize('C,H,W,B,groups,training', [(32, 16, 16, 2, 2, True), (64, 8, 12, 3, 1, True), (12, 6, 10, 1, 2, True),
                                                     (128, 8, 8, 2, 2, False), (32, 16, 16, 2, 1, False)])
def test_bn_backward_with_pool_gradient(C, H, W, B, groups, training):
    """pp_bn_lrelu_bwd_pool / pp_bn_lrelu_bwd_eval_pool: BatchNorm + LeakyReLU backward with the gradient of the following
    nn.MaxPool2d(2, 2) folded in, against torch autograd in fp64 of  loss = <y, dskip> + <max_pool2d(y), dpool>  (models/unet.py:
    109,123-127: the output of an encoder stage feeds the skip connection and the pooling)."""
    import torch.nn.functional as F
    from tests.test_gpu_ops import _lib, dev, nchw, nhwc, rel
    lib, st = _lib()
    g = torch.Generator().manual_seed(C + H + 7)
    N = B * groups
    z = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    gamma[0] = -0.7                               # negative scale: the window maximum of y is the MINIMUM of z there
    beta = torch.randn(C, generator=g)
    rm0 = torch.randn(C, generator=g) * 0.1
    rv0 = torch.rand(C, generator=g) + 0.5
    dy = torch.randn(N, C, H, W, generator=g)
    dp = torch.randn(N, C, H // 2, W // 2, generator=g)
    zr = z.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    ys = [F.leaky_relu(F.batch_norm(zr[gi * B:(gi + 1) * B], rm0.clone().double(), rv0.clone().double(), gr, br, training, 0.1, 1e-5), 0.01)
          for gi in range(groups)]
    yr = torch.cat(ys)
    ((yr * dy.double()).sum() + (F.max_pool2d(yr, 2, 2) * dp.double()).sum()).backward()

    ld = C + 4
    zd = torch.zeros(N, H, W, ld, device=dev()); zd[..., :C] = nhwc(z).to(dev())
    coef = torch.empty(4, groups, C, device=dev())
    rmd, rvd = rm0.to(dev()), rv0.to(dev())
    nbt = torch.zeros((), dtype=torch.int64, device=dev())
    ppg = B * H * W
    nws = lib.pp_bn_workspace(C, ppg, groups) + 12 * groups * C
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    gd, bd = gamma.to(dev()), beta.to(dev())
    mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
    dyd = torch.zeros(N, H, W, ld, device=dev()); dyd[..., :C] = nhwc(dy).to(dev())
    dpd = torch.zeros(N, H // 2, W // 2, ld, device=dev()); dpd[..., :C] = nhwc(dp).to(dev())
    dzd = torch.full((N, H, W, ld), 5.0, device=dev())
    dg, db, dbc = (torch.full((C,), 9.0, device=dev()) for _ in range(3))
    amax = torch.full((1,), -1.0, device=dev())
    if training:
        lib.pp_bn_train_stats(zd.data_ptr(), ld, C, ppg, groups, 1e-5, 0.1, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(),
                              rvd.data_ptr(), nbt.data_ptr(), mean, invstd, scale, shift, ws.data_ptr(), nws, st)
        lib.pp_bn_lrelu_bwd_pool(dyd.data_ptr(), ld, dpd.data_ptr(), ld, zd.data_ptr(), ld, scale, shift, mean, invstd, gd.data_ptr(),
                                 1, dzd.data_ptr(), ld, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, N, H, W, groups, 0.01,
                                 ws.data_ptr(), nws, amax.data_ptr(), st)
    else:
        assert groups == 1 or True
        lib.pp_bn_eval_coeffs(C, groups, 1e-5, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), mean, invstd, scale, shift, st)
        yd = torch.zeros(N, H, W, ld, device=dev())
        lib.pp_bn_lrelu_fwd(zd.data_ptr(), ld, scale, shift, yd.data_ptr(), ld, C, ppg, groups, 0.01, st)
        lib.pp_bn_lrelu_bwd_eval_pool(dyd.data_ptr(), ld, dpd.data_ptr(), ld, yd.data_ptr(), ld, scale, gd.data_ptr(), bd.data_ptr(),
                                      dzd.data_ptr(), ld, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, N, H, W, 0.01,
                                      ws.data_ptr(), nws, amax.data_ptr(), st)
    assert rel(nchw(dzd[..., :C]), zr.grad) < 1e-4
    assert torch.all(dzd[..., C:] == 5.0)
    assert rel(dg, gr.grad) < 1e-4 and rel(db, br.grad) < 1e-4
    assert abs(float(amax) - float(dzd[..., :C].abs().max())) <= 1e-6 * float(amax)
    if not training:
        ref_dbias = zr.grad.sum((0, 2, 3))
        assert float((dbc.cpu().double() - ref_dbias).abs().max()) < 1e-4 * float(zr.grad.abs().sum((0, 2, 3)).max())


@pytest.mark.parametrize('training', [True, False])
def test_pool_gradient_fused_matches_separate_pass(training):
    """The engine with the max-pool gradient folded into the BatchNorm backward (default) against the separate
    pp_maxpool2_bwd pass: one full-flags step, train- and eval-mode BN -- same sums in the same order up to the association
    of ONE addition per element (dskip + dpool before or after the window is chosen)."""
    from pacingpseudo_amd import engine as E
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags(init_ch=8, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    batch = O.synthetic_batch(3, 64, 96, seed=11, keep=0.05)
    saved = (E.FUSE_POOL_BWD, E.FUSE_POOL_FWD)
    out = {}
    try:
        for flag in (False, True):
            E.FUSE_POOL_BWD = E.FUSE_POOL_FWD = flag       # forward: pp_bn_lrelu_fwd_pool (train mode) instead of apply + max-pool
            torch.manual_seed(4)
            model = build_model(args)
            if not training:
                model.eval()
            opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
            out[flag] = iteration(model, opt, batch, args, 0 if training else 1)
    finally:
        E.FUSE_POOL_BWD, E.FUSE_POOL_FWD = saved
    for k, v in out[False][0].items():
        assert torch.equal(out[True][0][k], v), k           # the forward is the same arithmetic: bit-identical outputs
    for k, v in out[False][1].items():
        if v is None or (training and G.is_bias_before_bn(k)):
            continue
        assert G.rel_err(out[True][1][k].double().cpu().numpy(), v.double().cpu().numpy()) < 2e-6, k


def test_inference_repacks_weights_only_when_they_changed():
    """A forward without gradients packs the kernel-side weight layouts only when a convolution weight may have changed since
    the plan last packed them (optimizer slab version, torch in-place version counters, addresses): validation / inference
    used to re-pack all 23 layers per forward.  Every way the weights can change must be seen."""
    from pacingpseudo_amd.models import UNet
    from pacingpseudo_amd.optim import FusedAdam
    torch.manual_seed(5)
    net = UNet(input_ch=1, init_ch=8, max_ch=64, num_classes=5, output_stride=8, elab_end_points=False).cuda().eval()
    x = torch.randn(2, 1, 64, 64, device='cuda')
    with torch.no_grad():
        a = net(x)['segmentation/logits'].clone()
        b = net(x)['segmentation/logits'].clone()
    assert torch.equal(a, b)
    key = net._engine.last_plan.packed_key
    assert key is not None and net._engine._weights_key() == key
    # 1. load_state_dict (torch in-place copy into the parameters)
    sd = {k: (v * 1.5 if k.endswith('conv.weight') else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    with torch.no_grad():
        c = net(x)['segmentation/logits'].clone()
    ref = UNet(input_ch=1, init_ch=8, max_ch=64, num_classes=5, output_stride=8, elab_end_points=False).cuda().eval()
    ref.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(c, ref(x)['segmentation/logits'])
    assert not torch.equal(c, a)
    # 2. a fused optimizer step (raw-pointer update of the flat slab)
    net.train()
    opt = FusedAdam(net.parameters(), lr=1e-2)
    out = net(x)['segmentation/logits']
    out.square().mean().backward()
    opt.step()
    net.eval()
    with torch.no_grad():
        d = net(x)['segmentation/logits'].clone()
        ref.load_state_dict(net.state_dict())
        assert torch.equal(d, ref(x)['segmentation/logits'])
    assert not torch.equal(d, c)
    # 3. a write through .data (EMA updates, collectives on a view of the slab): no counter sees it -- the caller announces it
    # with invalidate_packed(), and the next forward without gradients packs again
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 4:
                p.data.mul_(0.5)
        net._engine.invalidate_packed()
        e = net(x)['segmentation/logits'].clone()
        ref.load_state_dict(net.state_dict())
        assert torch.equal(e, ref(x)['segmentation/logits'])
    assert not torch.equal(e, d)
    # ... and load_state_dict announces itself (post-hook): the packed key of every plan is forgotten
    net.load_state_dict(net.state_dict())
    assert all(p.packed_key is None for p in net._engine.plans.values())


def test_second_stream_weight_gradients_are_bit_identical():
    """Weight gradients on a second HIP stream (engine.WGRAD_STREAM: two dz buffers in turn, own workspace, joined before the
    optimizer) against the single-stream order: the same kernels on the same data, so every gradient must be bit-identical --
    anything else is a race (a dz buffer overwritten too early, a shared workspace).  Round 5: on the second stream the direct
    weight-gradient kernels get a CU budget (engine.WGRAD_CUS_SIDE, 192 of 256), i.e. another number of tile walkers and so
    another partition of the same fixed-order sums: bit-identity is asserted with the SAME budget in both runs, and the default
    budgets (192 beside 256) are held to fp32 summation-order differences."""
    from pacingpseudo_amd import engine as E
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags()
    batch = O.synthetic_batch(2, 128, 128, seed=3, keep=0.05)
    saved = (E.WGRAD_STREAM, E.WGRAD_CUS_FULL, E.WGRAD_CUS_SIDE)
    runs = {}
    try:
        for tag, flag, full in (('one', False, E.WGRAD_CUS_SIDE), ('two', True, E.WGRAD_CUS_SIDE), ('one_default', False, saved[1])):
            E.WGRAD_STREAM, E.WGRAD_CUS_FULL = flag, full
            torch.manual_seed(1)
            model = build_model(args)
            opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
            steps = 3 if tag != 'one_default' else 1
            for _ in range(steps):                   # three steps: the buffers and events are reused across steps
                rec, grads = iteration(model, opt, batch, args, 0)
            runs[tag] = (rec, grads, {k: v.detach().clone() for k, v in model.state_dict().items()})
            assert (model.engine._wg_stream is not None) == flag
    finally:
        E.WGRAD_STREAM, E.WGRAD_CUS_FULL, E.WGRAD_CUS_SIDE = saved
    for k, v in runs['one'][1].items():
        if v is not None:
            assert torch.equal(runs['two'][1][k], v), k
    for k, v in runs['one'][2].items():
        assert torch.equal(runs['two'][2][k], v), k
    # the default budgets: one step from the same initial weights, first-step gradients of the 192-CU run recomputed
    torch.manual_seed(1)
    E.WGRAD_STREAM = True
    try:
        model = build_model(args)
        opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
        _, g192 = iteration(model, opt, batch, args, 0)
    finally:
        E.WGRAD_STREAM = saved[0]
    for k, v in runs['one_default'][1].items():
        if v is not None and not G.is_bias_before_bn(k):
            assert G.rel_err(g192[k].double().cpu().numpy(), v.double().cpu().numpy()) < 2e-6, k


def _lazy_rows(groups, C, ld, g):
    """(groups, 3, ld) coefficient rows: random scale (some negative) / shift, slope 0.01; identity beyond channel C."""
    coef = torch.zeros(groups, 3, ld)
    coef[:, 0] = 1.0
    coef[:, 2] = 1.0
    coef[:, 0, :C] = torch.randn(groups, C, generator=g) * 0.7 + 0.3
    coef[:, 1, :C] = torch.randn(groups, C, generator=g) * 0.5
    coef[:, 2, :C] = 0.01
    return coef


def _lazy_ref(z_nchw, coef, groups, C):
    """y = lrelu(z * scale + shift) per statistics group, in double."""
    N = z_nchw.shape[0]
    per = N // groups
    out = torch.empty_like(z_nchw, dtype=torch.float64)
    for gi in range(groups):
        sc, sh, sl = (coef[gi, r, :C].double().view(1, C, 1, 1) for r in range(3))
        v = z_nchw[gi * per:(gi + 1) * per].double() * sc + sh
        out[gi * per:(gi + 1) * per] = torch.where(v > 0, v, v * sl)
    return out


@pytest.mark.parametrize('C,N,H,W,groups', [(32, 4, 16, 16, 2), (12, 2, 6, 10, 1), (64, 6, 8, 4, 2)])
def test_lazy_entry_points_against_torch(C, N, H, W, groups):
    """Every *_lazy entry point of include/pacingpseudo_hip.h on a lazy tensor (z + coefficient rows) against the ordinary
    entry point's reference applied to y = lrelu(z * scale + shift) computed in fp64: pp_lazy_materialize,
    pp_conv1x1_nhwc_to_nchw_fwd_lazy / _bwd_lazy (the halo-kernel forms: test_halo_kernels_with_lazy_input below).  The tensor is a channel SLICE of a wider
    buffer (coef + c0, ld kept), as the engine passes the halves of a concatenation buffer."""
    import ctypes
    import torch.nn.functional as F
    from pacingpseudo_amd._lib import PpLazyIn
    from tests.test_gpu_ops import _lib, dev, nchw, nhwc, rel
    lib, st = _lib()
    g = torch.Generator().manual_seed(C * 7 + N)
    c0, ld = 8, C + 16                                  # the view starts at channel 8 of a (C + 16)-channel buffer
    z = torch.randn(N, C, H, W, generator=g)
    coef_all = _lazy_rows(groups, ld, ld, g)
    coef = coef_all[:, :, c0:c0 + C].contiguous()       # rows of the view's channels (reference side)
    buf = torch.full((N, H, W, ld), 3.0, device=dev())
    buf[..., c0:c0 + C] = nhwc(z).to(dev())
    cd = coef_all.to(dev()).contiguous()
    view_ptr = buf.data_ptr() + 4 * c0
    lz = PpLazyIn(cd.data_ptr() + 4 * c0, ld, groups)
    y = _lazy_ref(z, coef, groups, C)

    out = torch.full((N, H, W, C + 4), 9.0, device=dev())
    lib.pp_lazy_materialize(view_ptr, ld, ctypes.byref(lz), out.data_ptr(), C + 4, C, N, H * W, st)
    assert rel(nchw(out[..., :C]), y) < 1e-6 and torch.all(out[..., C:] == 9.0)

    K = 5
    w = torch.randn(K, C, generator=g) / C ** 0.5
    b = torch.randn(K, generator=g)
    logits = torch.empty(N, K, H, W, device=dev())
    wd, bd = w.to(dev()), b.to(dev())                  # (kept alive: a temporary's memory is recycled as soon as it is dropped)
    lib.pp_conv1x1_nhwc_to_nchw_fwd_lazy(view_ptr, ld, C, wd.data_ptr(), bd.data_ptr(), logits.data_ptr(), K, N,
                                         H * W, ctypes.byref(lz), st)
    yr2 = y.clone().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    lr_ = F.conv2d(yr2, wr.view(K, C, 1, 1), br)
    assert rel(logits, lr_) < 1e-5
    dl = torch.randn(N, K, H, W, generator=g)
    lr_.backward(dl.double())
    nws = lib.pp_conv1x1_bwd_workspace(K, C, N, H * W)
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    dxh = torch.zeros(N, H, W, C, device=dev())
    dw, db = torch.zeros(K, C, device=dev()), torch.zeros(K, device=dev())
    dld = dl.to(dev())
    lib.pp_conv1x1_nchw_to_nhwc_bwd_lazy(dld.data_ptr(), view_ptr, ld, C, wd.data_ptr(), dxh.data_ptr(), C,
                                         dw.data_ptr(), db.data_ptr(), K, N, H * W, 0, 0, ws.data_ptr(), nws, ctypes.byref(lz), st)
    assert rel(nchw(dxh), yr2.grad) < 1e-5 and rel(dw, wr.grad) < 1e-5 and rel(db, br.grad) < 1e-5
    assert torch.all(buf[..., :c0] == 3.0) and torch.all(buf[..., c0 + C:] == 3.0)


@pytest.mark.parametrize('training,do_memory', [(True, True), (False, True), (True, False)])
def test_aux_path_forward_stand_alone(training, do_memory):
    """AuxPath.forward(end_points, scribble, step) as the reference exposes it (models/aux_path_memory.py:46-66), called by
    itself: outputs, the memory bank after its update, and the gradients into the end points and every parameter, against the
    oracle's aux_forward (two calls: the second one meets a non-empty bank -> the cosine-similarity ensemble branch)."""
    from pacingpseudo_amd.models.aux_path_memory import AuxPath
    K, hid, B, h, w, H, W = 5, 16, 2, 8, 8, 64, 64
    args = O.full_flags(hid_ch=hid, feat_ch=[32, 32], do_memory=do_memory)
    torch.manual_seed(9)
    aux = AuxPath(num_classes=K, feat_stage=args.feat_stage, feat_ch=args.feat_ch, hid_ch=hid, aux_drop_prob=0.0,
                  do_memory=do_memory, max_step=args.epoch, update_momentum=args.update_momentum,
                  ensemble_mode=args.ensemble_mode).cuda()
    aux.train(training)
    with torch.no_grad():
        aux.layer_bottleneck[2].running_mean.normal_(0, 0.1)
        aux.layer_bottleneck[2].running_var.uniform_(0.5, 1.5)
    sd = {'aux_path.' + k: v.detach().cpu().clone() for k, v in aux.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    batch = O.synthetic_batch(B, H, W, seed=6, keep=0.05)
    scribble = batch['scribble']
    for call in range(2):
        feats = {s: torch.randn(B, 32, h, w, generator=g) for s in args.feat_stage}
        wa, wm = torch.randn(B, K, H, W, generator=g), torch.randn(K, K, 1, 1, generator=g)
        # oracle
        ref_in = {s: v.clone().requires_grad_(True) for s, v in feats.items()}
        keys = [k for k in sd if sd[k].is_floating_point() and not k.endswith(('running_mean', 'running_var', 'memory_bank'))]
        for k in keys:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        ro = O.aux_forward(sd, ref_in, scribble, 3, args, training)
        loss_r = (ro['logits_aux_cls'] * wa).sum() + ((ro['logits_memory'] * wm).sum() if do_memory else 0.0)
        loss_r.backward()
        # device
        dev_in = {s: v.cuda().requires_grad_(True) for s, v in feats.items()}
        aux.zero_grad()
        do = aux(dev_in, scribble.cuda(), 3)
        assert sorted(do) == sorted(k for k in ro if k != 'aux_features')
        loss_d = (do['logits_aux_cls'] * wa.cuda()).sum() + ((do['logits_memory'] * wm.cuda()).sum() if do_memory else 0.0)
        loss_d.backward()
        assert G.rel_err(do['logits_aux_cls'].detach().cpu().numpy(), ro['logits_aux_cls'].detach().numpy()) < TOL_OUT
        assert torch.equal(do['aux_targets'].cpu(), ro['aux_targets'])
        if do_memory:
            assert G.rel_err(do['logits_memory'].detach().cpu().numpy(), ro['logits_memory'].detach().numpy()) < TOL_OUT
            assert torch.equal(do['memory_target'].cpu(), ro['memory_target'])
            assert G.rel_err(aux.memory_bank.detach().cpu().numpy(), sd['aux_path.memory_bank'].numpy()) < TOL_OUT, call
        for s in args.feat_stage:
            assert G.rel_err(dev_in[s].grad.cpu().numpy(), ref_in[s].grad.numpy()) < TOL_GRAD, (call, s)
        names = dict(aux.named_parameters())
        for k in keys:
            n = k[len('aux_path.'):]
            got, want = names[n].grad, sd[k].grad
            if training and n == 'layer_bottleneck.1.bias':          # conv bias in front of train-mode BN: exactly zero
                assert float(got.abs().max()) < 2e-5
                continue
            assert G.rel_err(got.cpu().numpy(), want.numpy()) < TOL_GRAD, (call, k)
        for k in keys:
            sd[k].requires_grad_(False)
        for k in ('running_mean', 'running_var'):
            assert G.rel_err(getattr(aux.layer_bottleneck[2], k).cpu().numpy(), sd['aux_path.layer_bottleneck.2.' + k].numpy()) < TOL_OUT


@pytest.mark.parametrize('B,H,W,Cin,Cout,groups,h16', [(2, 8, 32, 32, 32, 2, False), (4, 8, 64, 64, 64, 2, False), (3, 12, 32, 64, 32, 1, False),
                                                       (2, 8, 32, 32, 64, 2, False), (2, 16, 32, 32, 32, 2, True), (2, 8, 64, 64, 64, 1, True)])
def test_direct_convolution_with_lazy_input(B, H, W, Cin, Cout, groups, h16):
    """pp_conv3x3_fwd_bn_lazy (two-half halo kernel: BatchNorm + LeakyReLU applied while the patch is staged, zero padding on y)
    and pp_conv3x3_bwd_weight_f16x3_lazy (halo-tile kernels, one pair and several pairs per block) against the ordinary entry
    points fed the materialised y (pp_lazy_materialize): bit-identical outputs, statistics and weight gradients -- the same
    arithmetic element for element.  h16: the `_h16` twins."""
    import ctypes
    import math
    from pacingpseudo_amd._lib import PpLazyIn, lib_for
    from tests.test_gpu_ops import dev
    K = lib_for(2 if h16 else 4)
    from pacingpseudo_amd._lib import lib, stream_ptr
    st = stream_ptr()
    dt = torch.float16 if h16 else torch.float32
    assert K.pp_conv3x3_lazy_ok(Cin, Cout, B, H, W, 1) == 1
    g = torch.Generator().manual_seed(B * 10 + Cin + Cout)
    z = (torch.randn(B, H, W, Cin, generator=g) * 1.3).to(dev()).to(dt)
    coef = _lazy_rows(groups, Cin, Cin, g).to(dev()).contiguous()
    lz = PpLazyIn(coef.data_ptr(), Cin, groups)
    y = torch.empty_like(z)
    K.pp_lazy_materialize(z.data_ptr(), Cin, ctypes.byref(lz), y.data_ptr(), Cin, Cin, B, H * W, st)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    wf = torch.empty(Cout, 9, Cin, device=dev())
    lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), None, st)
    nst = lib.pp_conv3x3_bn_stats_bytes(Cout, B, H, W, groups)
    outs = []
    for lazy in (False, True):
        out = torch.zeros(B, H, W, Cout, device=dev(), dtype=dt)
        stats = torch.zeros(nst // 8 + 2, device=dev(), dtype=torch.float64)
        rows = ctypes.c_int(0)
        a = ((z if lazy else y).data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, H, W, 1, 1, None, 1,
             None, None, 0.01, groups, stats.data_ptr(), nst, ctypes.byref(rows))
        if lazy:
            K.pp_conv3x3_fwd_bn_lazy(*a, ctypes.byref(lz), st)
        else:
            K.pp_conv3x3_fwd_bn(*a, st)
        torch.cuda.synchronize()
        outs.append((out, stats[:groups * rows.value * 2 * Cout].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dz = (torch.randn(B, H, W, Cout, generator=g) * 3e-3).to(dev()).to(dt)
    amax = dz.float().abs().max().reshape(1).contiguous()
    nws = lib.pp_conv3x3_bwd_weight_workspace(Cout, Cin, B, H, W)
    dws = []
    for lazy in (False, True):
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev())
        a = (dz.data_ptr(), Cout, Cout, (z if lazy else y).data_ptr(), Cin, Cin, Cin, B, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nws,
             amax.data_ptr())
        if lazy:
            K.pp_conv3x3_bwd_weight_f16x3_lazy(*a, ctypes.byref(lz), st)
        else:
            K.pp_conv3x3_bwd_weight_f16x3(*a, st)
        torch.cuda.synchronize()
        dws.append(dw)
    assert torch.equal(dws[0], dws[1])
    # a shape without the two-half halo kernel says so instead of ignoring the coefficients
    assert K.pp_conv3x3_lazy_ok(128, 128, B, H, W, 1) == 0 and K.pp_conv3x3_lazy_ok(Cin, Cout, B, H, 24, 1) == 0


def test_auxiliary_path_on_the_second_stream_is_bit_identical():
    """Round 5: the auxiliary path's forward (bottleneck conv + BatchNorm, classifier, its partial CE, the bank update) runs on the second
    HIP stream beside the decoder's forward pass -- forked when the encoder is enqueued, joined behind the segmentation losses, with
    its own workspace and statistics rows (engine.AUX_SIDE) -- and so does the head of its backward (partial CE, classifier, bank CE),
    beside the 1x1 head and the first decoder stages.  Same kernels on the same data: three steps with it must equal three
    steps without it bit for bit (outputs, every gradient, BatchNorm buffers, memory bank) -- anything else is a race."""
    from pacingpseudo_amd import engine as E
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags()
    batch = O.synthetic_batch(2, 128, 128, seed=5, keep=0.05)
    saved = E.AUX_SIDE
    runs = {}
    try:
        for flag in (False, True):
            E.AUX_SIDE = flag
            torch.manual_seed(1)
            model = build_model(args)
            opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
            for _ in range(3):
                rec, grads = iteration(model, opt, batch, args, 0)
            runs[flag] = (rec, grads, {k: v.detach().clone() for k, v in model.state_dict().items()})
    finally:
        E.AUX_SIDE = saved
    for k, v in runs[False][0].items():
        assert torch.equal(runs[True][0][k], v), k
    for k, v in runs[False][1].items():
        if v is not None:
            assert torch.equal(runs[True][1][k], v), k
    for k, v in runs[False][2].items():
        assert torch.equal(runs[True][2][k], v), k
