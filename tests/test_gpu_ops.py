"""Per-kernel parity tests: every C-ABI entry point against the reference arithmetic (PyTorch fp32/fp64 on the
CPU, i.e. the same aten ops the reference dispatches, and the oracle's restatements for the loss / memory code).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402

TOL = 1e-4          # north_star: outputs within 1e-4 fp32 of the reference CPU path


def _lib():
    from pacingpseudo_amd._lib import lib, stream_ptr
    return lib, stream_ptr()


def dev():
    return torch.device('cuda', 0)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def pack_w(w, ipad):
    lib, st = _lib()
    O_, I = w.shape[:2]
    wf = torch.empty(O_, 9, ipad, device=dev())
    wb = torch.empty(I, 9, O_, device=dev())
    lib.pp_pack_conv3x3_weights(w.data_ptr(), O_, I, ipad, wf.data_ptr(), wb.data_ptr(), st)
    return wf, wb


CONV_CASES = [
    # B, H, W, Cin, Cout, dil
    (2, 16, 16, 1, 32, 1),          # enc1.c1: single input channel padded to 4
    (2, 32, 32, 32, 32, 1),
    (1, 32, 32, 64, 64, 2),
    (2, 32, 32, 128, 128, 4),
    (1, 64, 64, 96, 32, 1),         # dec1.c1 shape class
    (2, 16, 16, 1024, 512, 1),      # dec5.c1 shape class (deep K)
    (1, 16, 16, 192, 64, 1),
    (2, 8, 8, 12, 20, 1),           # ragged: channels not multiples of 32, tiny image
    (3, 10, 6, 8, 4, 2),            # ragged pixel count (not a multiple of any tile), non-square
    (2, 16, 128, 64, 64, 1),        # wide rows: tap-fused weight-gradient kernel, 2x2 tiles of 32
    (1, 8, 64, 1, 32, 1),           # ... with the padded single-channel input
    (2, 8, 192, 20, 12, 1),         # ... ragged channel tiles
    (6, 128, 128, 32, 32, 1),       # persistent halo-tile kernel: more tiles (768) than blocks, weights resident
    (5, 128, 64, 64, 32, 1),        # ... two channel chunks forward, two output-channel groups in the data gradient
    (2, 12, 96, 32, 64, 1),         # ... 3 x 3 tiles per image, image borders inside every tile column
    (2, 12, 20, 3, 48, 2),          # first-layer kernels (input padded to 4 channels): dilated, 3 real channels, 48 outputs
    (3, 8, 8, 2, 16, 1),            # ... one 16-row MFMA tile, tiny image
    (2, 64, 64, 1, 32, 1),          # ... several waves and blocks of partial sums
    (3, 12, 32, 3, 48, 2),          # ... MFMA forward kernel, dilated, odd number of 16-pixel groups per wave
]


@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', CONV_CASES)
def test_conv3x3_fwd_bwd(B, H, W, Cin, Cout, dil):
    lib, st = _lib()
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), 1, dil, dil)
    yr.backward(dy.double())

    ipad = (Cin + 3) // 4 * 4
    ld_in = ipad + 8                 # exercise leading dimensions (channel slices of wider buffers)
    ld_out = Cout + 4 if Cout % 4 == 0 else Cout + 3
    ld_out = (ld_out + 3) // 4 * 4
    xin = torch.zeros(B, H, W, ld_in, device=dev())
    xin[..., :Cin] = nhwc(x).to(dev())
    wd = w.to(dev())
    wf, wb = pack_w(wd, ipad)
    out = torch.full((B, H, W, ld_out), 7.0, device=dev())
    lib.pp_conv3x3_fwd(xin.data_ptr(), ld_in, ipad, wf.data_ptr(), b.to(dev()).data_ptr(), out.data_ptr(), ld_out, Cout,
                       B, H, W, dil, 0, st)
    torch.cuda.synchronize()
    assert rel(nchw(out[..., :Cout]), yr) < TOL
    assert torch.all(out[..., Cout:] == 7.0), 'kernel wrote outside its channel slice'
    # accumulate
    lib.pp_conv3x3_fwd(xin.data_ptr(), ld_in, ipad, wf.data_ptr(), None, out.data_ptr(), ld_out, Cout, B, H, W, dil, 1, st)
    yr2 = 2 * yr - b.double().view(1, -1, 1, 1)
    assert rel(nchw(out[..., :Cout]), yr2) < TOL

    # data gradient (needs Cin % 4 == 0: the first layer never asks for it)
    dz = torch.zeros(B, H, W, ld_out, device=dev())
    dz[..., :Cout] = nhwc(dy).to(dev())
    if Cin % 4 == 0 and Cout % 4 == 0:
        dx = torch.full((B, H, W, ld_in), 3.0, device=dev())
        lib.pp_conv3x3_bwd_data(dz.data_ptr(), ld_out, Cout, wb.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W, dil, 0, st)
        assert rel(nchw(dx[..., :Cin]), xr.grad) < TOL
        assert torch.all(dx[..., Cin:] == 3.0)
        lib.pp_conv3x3_bwd_data(dz.data_ptr(), ld_out, Cout, wb.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W, dil, 1, st)
        assert rel(nchw(dx[..., :Cin]), 2 * xr.grad) < TOL
    # weight gradient
    if Cout % 4 == 0:
        nbytes = lib.pp_conv3x3_bwd_weight_workspace(Cout, ipad, B, H, W)
        ws = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev())
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev())
        lib.pp_conv3x3_bwd_weight(dz.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, ipad, Cin, B, H, W, dil,
                                  dw.data_ptr(), 0, ws.data_ptr(), nbytes, st)
        assert rel(dw, wr.grad) < TOL
        lib.pp_conv3x3_bwd_weight(dz.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, ipad, Cin, B, H, W, dil,
                                  dw.data_ptr(), 1, ws.data_ptr(), nbytes, st)
        assert rel(dw, 2 * wr.grad) < TOL


def test_conv_identity_asymmetric():
    """A = delta kernel with an asymmetric tap: catches swapped row/col or flipped-tap mistakes exactly."""
    lib, st = _lib()
    B, H, W, C = 1, 8, 8, 32
    x = torch.arange(B * H * W * C, dtype=torch.float32).reshape(B, H, W, C).to(dev())
    w = torch.zeros(C, C, 3, 3)
    for c in range(C):
        w[c, (c * 7 + 3) % C, 0, 2] = 1.0          # out[c] = in[(7c+3)%C] shifted by (dy=-1, dx=+1)
    wf, _ = pack_w(w.to(dev()), C)
    out = torch.empty(B, H, W, C, device=dev())
    lib.pp_conv3x3_fwd(x.data_ptr(), C, C, wf.data_ptr(), None, out.data_ptr(), C, C, B, H, W, 1, 0, st)
    ref = F.conv2d(nchw(x.cpu()), w, None, 1, 1, 1)
    assert torch.equal(nchw(out.cpu()), ref)


@pytest.mark.parametrize('C,H,W,B,groups,training', [(32, 16, 16, 2, 2, True), (64, 8, 8, 3, 1, True),
                                                     (12, 6, 10, 1, 2, True), (1024, 4, 4, 2, 2, True),
                                                     (32, 16, 16, 2, 2, False), (96, 32, 32, 2, 1, True)])
def test_bn_lrelu(C, H, W, B, groups, training):
    lib, st = _lib()
    g = torch.Generator().manual_seed(C + H)
    N = B * groups
    z = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    gamma[0] = -0.7                               # negative scale: sign handling of the leaky slope
    beta = torch.randn(C, generator=g)
    rm0 = torch.randn(C, generator=g) * 0.1
    rv0 = torch.rand(C, generator=g) + 0.5
    dy = torch.randn(N, C, H, W, generator=g)
    # reference: one module call per group, in order
    rm, rv = rm0.clone().double(), rv0.clone().double()
    zr = z.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    ys = []
    for gi in range(groups):
        ys.append(F.leaky_relu(F.batch_norm(zr[gi * B:(gi + 1) * B], rm, rv, gr, br, training, 0.1, 1e-5), 0.01))
    yr = torch.cat(ys)
    yr.backward(dy.double())

    ld = C + 4
    zd = torch.zeros(N, H, W, ld, device=dev()); zd[..., :C] = nhwc(z).to(dev())
    yd = torch.zeros(N, H, W, ld, device=dev())
    coef = torch.empty(4, groups, C, device=dev())
    rmd, rvd = rm0.to(dev()), rv0.to(dev())
    nbt = torch.zeros((), dtype=torch.int64, device=dev())
    ppg = B * H * W
    nws = lib.pp_bn_workspace(C, ppg, groups) + 12 * groups * C
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    gd, bd = gamma.to(dev()), beta.to(dev())
    mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
    if training:
        lib.pp_bn_train_stats(zd.data_ptr(), ld, C, ppg, groups, 1e-5, 0.1, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(),
                              rvd.data_ptr(), nbt.data_ptr(), mean, invstd, scale, shift, ws.data_ptr(), nws, st)
        assert int(nbt) == groups
        assert rel(rmd, rm) < 1e-5 and rel(rvd, rv) < 1e-5
    else:
        lib.pp_bn_eval_coeffs(C, groups, 1e-5, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), mean,
                              invstd, scale, shift, st)
    lib.pp_bn_lrelu_fwd(zd.data_ptr(), ld, scale, shift, yd.data_ptr(), ld, C, ppg, groups, 0.01, st)
    assert rel(nchw(yd[..., :C]), yr) < TOL
    dyd = torch.zeros(N, H, W, ld, device=dev()); dyd[..., :C] = nhwc(dy).to(dev())
    dzd = torch.zeros(N, H, W, ld, device=dev())
    dg, db, dbc = (torch.full((C,), 9.0, device=dev()) for _ in range(3))
    lib.pp_bn_lrelu_bwd(dyd.data_ptr(), ld, zd.data_ptr(), ld, scale, shift, mean, invstd, gd.data_ptr(),
                        1 if training else 0, dzd.data_ptr(), ld, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, ppg,
                        groups, 0.01, ws.data_ptr(), nws, st)
    assert rel(nchw(dzd[..., :C]), zr.grad) < TOL
    assert rel(dg, gr.grad) < TOL and rel(db, br.grad) < TOL
    ref_dbias = zr.grad.sum((0, 2, 3))
    assert float((dbc.cpu().double() - ref_dbias).abs().max()) < TOL * float(zr.grad.abs().sum((0, 2, 3)).max())


@pytest.mark.parametrize('C,N,H,W', [(32, 2, 16, 16), (12, 1, 6, 10), (64, 3, 8, 4)])
def test_maxpool(C, N, H, W):
    lib, st = _lib()
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, H, W, generator=g)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 5.0                       # a tie: the first maximum must take the gradient
    dy = torch.randn(N, C, H // 2, W // 2, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    yr.backward(dy)
    xd = nhwc(x).to(dev())
    yd = torch.empty(N, H // 2, W // 2, C, device=dev())
    lib.pp_maxpool2_fwd(xd.data_ptr(), C, yd.data_ptr(), C, C, N, H, W, st)
    assert torch.equal(nchw(yd).cpu(), yr.detach())
    dxd = torch.ones(N, H, W, C, device=dev())
    lib.pp_maxpool2_bwd(xd.data_ptr(), C, nhwc(dy).to(dev()).data_ptr(), C, dxd.data_ptr(), C, C, N, H, W, 1, st)
    assert torch.equal(nchw(dxd).cpu(), xr.grad + 1.0)


@pytest.mark.parametrize('C,N,Hi,Wi,Ho,Wo', [(32, 2, 8, 8, 16, 16), (8, 1, 32, 32, 256, 256), (4, 2, 5, 7, 10, 14),
                                             (16, 1, 8, 8, 8, 8), (64, 1, 28, 28, 224, 224),
                                             (8, 3, 7, 9, 14, 18), (12, 1, 1, 6, 2, 12), (64, 2, 32, 32, 64, 64)])   # odd sizes: the 2 x 2-block backward's edge blocks
def test_bilinear(C, N, Hi, Wi, Ho, Wo):
    lib, st = _lib()
    g = torch.Generator().manual_seed(Hi * Wo)
    x = torch.randn(N, C, Hi, Wi, generator=g)
    dy = torch.randn(N, C, Ho, Wo, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, size=(Ho, Wo), mode='bilinear', align_corners=True)
    yr.backward(dy.double())
    xd = nhwc(x).to(dev())
    yd = torch.empty(N, Ho, Wo, C, device=dev())
    lib.pp_bilinear_fwd(xd.data_ptr(), C, yd.data_ptr(), C, C, N, Hi, Wi, Ho, Wo, st)
    assert rel(nchw(yd), yr) < 1e-5
    if (Hi, Wi) == (Ho, Wo):
        assert torch.equal(yd, xd), 'scale factor 1 must be an exact identity'
    dxd = torch.empty(N, Hi, Wi, C, device=dev())
    lib.pp_bilinear_bwd(nhwc(dy).to(dev()).data_ptr(), C, dxd.data_ptr(), C, C, N, Hi, Wi, Ho, Wo, 0, st)
    assert rel(nchw(dxd), xr.grad) < 1e-5


@pytest.mark.parametrize('C,K,N,H,W,bias', [(32, 5, 2, 16, 16, True), (64, 5, 1, 8, 8, False), (8, 4, 3, 5, 7, True),
                                            (32, 5, 2, 64, 64, True),      # streaming kernels, blocks inside one image (unrolled path)
                                            (128, 8, 1, 32, 32, True), (16, 2, 3, 24, 40, True), (12, 3, 2, 9, 9, True)])
def test_conv1x1_head(C, K, N, H, W, bias):
    lib, st = _lib()
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 1, 1, generator=g) / math.sqrt(C)
    b = torch.randn(K, generator=g) if bias else None
    dl = torch.randn(N, K, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br)
    yr.backward(dl.double())
    xd = nhwc(x).to(dev())
    wd = w.to(dev())
    logits = torch.empty(N, K, H, W, device=dev())
    lib.pp_conv1x1_nhwc_to_nchw_fwd(xd.data_ptr(), C, C, wd.data_ptr(), b.to(dev()).data_ptr() if bias else None,
                                    logits.data_ptr(), K, N, H * W, st)
    assert rel(logits, yr) < 1e-5
    nws = lib.pp_conv1x1_bwd_workspace(K, C, N, H * W)
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    dx = torch.empty(N, H, W, C, device=dev())
    dw = torch.empty(K, C, device=dev())
    db = torch.empty(K, device=dev())
    lib.pp_conv1x1_nchw_to_nhwc_bwd(dl.to(dev()).data_ptr(), xd.data_ptr(), C, C, wd.data_ptr(), dx.data_ptr(), C,
                                    dw.data_ptr(), db.data_ptr() if bias else None, K, N, H * W, 0, 0, ws.data_ptr(), nws, st)
    assert rel(nchw(dx), xr.grad) < 1e-5
    assert rel(dw, wr.grad.view(K, C)) < 1e-5
    if bias:
        assert rel(db, br.grad) < 1e-5


def _loss_inputs(N, K, H, W, seed, all_ignored=False):
    g = torch.Generator().manual_seed(seed)
    zw = torch.randn(N, K, H, W, generator=g) * 2
    zs = torch.randn(N, K, H, W, generator=g) * 2
    t = torch.randint(0, K + 1, (N, H, W), generator=g)
    if all_ignored:
        t[:] = K
    scb = F.one_hot(t, K + 1).permute(0, 3, 1, 2).float().contiguous()
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.3).float()
    return zw, zs, t, scb, mask


def test_argmax_bit_exact():
    lib, st = _lib()
    zw, _, t, scb, _ = _loss_inputs(3, 5, 17, 9, 0)
    out = torch.empty(3, 17, 9, dtype=torch.int64, device=dev())
    lib.pp_argmax_channels(scb.to(dev()).data_ptr(), 3, 6, 17 * 9, out.data_ptr(), st)
    assert torch.equal(out.cpu(), scb.argmax(1)) and torch.equal(out.cpu(), t)
    zw[0, 1, 0, 0] = zw[0, 3, 0, 0] = 9.0                     # tie -> first index
    lib.pp_argmax_channels(zw.to(dev()).data_ptr(), 3, 5, 17 * 9, out.data_ptr(), st)
    assert torch.equal(out.cpu(), zw.argmax(1))


@pytest.mark.parametrize('variant,use_mask,detach', [('ce_loss', True, False), ('ce_loss', False, False),
                                                     ('ce_loss', True, True), ('l1_loss', True, False),
                                                     ('l2_loss', False, True), ('kl_loss', True, True),
                                                     (None, True, False)])
@pytest.mark.parametrize('spread', [1.0, 60.0])
def test_seg_losses(variant, use_mask, detach, spread):
    """spread 60: logit gaps beyond 87, where exp() underflows and probabilities are exactly 0 in fp32 -- the confident
    late-training regime.  Gradients must stay finite and equal torch's (r02: the quotient form of the ce/kl consistency
    gradient produced 0 * inf = NaN there and killed 3 of 10 long training runs)."""
    lib, st = _lib()
    N, K, H, W = 2, 5, 24, 20
    zw, zs, t, scb, mask = _loss_inputs(N, K, H, W, 3)
    zw, zs = zw * spread, zs * spread
    gw = dict(pce=0.7, ent=0.3, cr=1.9)
    zwr, zsr = zw.double().requires_grad_(True), zs.double().requires_grad_(True)
    m = mask.double() if use_mask else None
    pce = O.partial_cross_entropy_loss(zwr, t, K)
    ent = O.entropy_minimization_loss(zwr, m)
    total = gw['pce'] * pce + gw['ent'] * ent
    cr = None
    if variant:
        pw = torch.softmax(zwr, 1)
        if detach:
            pw = pw.detach()
        cr = {'ce_loss': lambda: O.soft_label_cross_entropy_loss(zsr, pw, m),
              'l1_loss': lambda: O.l1_loss(torch.softmax(zsr, 1), pw, m),
              'l2_loss': lambda: O.l2_loss(torch.softmax(zsr, 1), pw, m),
              'kl_loss': lambda: O.kl_loss(zsr, zwr, m)}[variant]()
        total = total + gw['cr'] * cr
    total.backward()

    vcode = {None: 0, 'ce_loss': 1, 'l1_loss': 2, 'l2_loss': 3, 'kl_loss': 4}[variant]
    zwd, zsd, td = zw.to(dev()), zs.to(dev()), t.to(dev())
    md = mask.to(dev()) if use_mask else None
    sums = torch.zeros(6, dtype=torch.float64, device=dev())
    nws = lib.pp_seg_losses_workspace(N, H * W)
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    lib.pp_seg_losses_fwd(zwd.data_ptr(), zsd.data_ptr() if variant else None, td.data_ptr(),
                          md.data_ptr() if use_mask else None, N, K, H * W, K, 1, vcode, sums.data_ptr(), ws.data_ptr(), nws, st)
    lp, le, lc = (torch.zeros((), device=dev()) for _ in range(3))
    lib.pp_losses_finalize(sums.data_ptr(), 1 if use_mask else 0, lp.data_ptr(), le.data_ptr(), lc.data_ptr() if variant else None, st)
    assert abs(float(lp) - float(pce)) < 1e-5 * max(1, abs(float(pce)))
    assert abs(float(le) - float(ent)) < 1e-5 * max(1, abs(float(ent)))
    if variant:
        assert abs(float(lc) - float(cr)) < 1e-5 * max(1, abs(float(cr)))
    dzw = torch.empty_like(zwd); dzs = torch.zeros_like(zsd)
    gs = {k: torch.tensor(v, device=dev()) for k, v in gw.items()}
    lib.pp_seg_losses_bwd(zwd.data_ptr(), zsd.data_ptr() if variant else None, td.data_ptr(),
                          md.data_ptr() if use_mask else None, N, K, H * W, K, 1, vcode, 1 if detach else 0, sums.data_ptr(),
                          gs['pce'].data_ptr(), gs['ent'].data_ptr(), gs['cr'].data_ptr(), 1.0, dzw.data_ptr(),
                          dzs.data_ptr() if variant else None, st)
    assert torch.isfinite(dzw).all() and torch.isfinite(dzs).all()
    assert rel(dzw, zwr.grad) < TOL
    if variant:
        assert rel(dzs, zsr.grad) < TOL


def test_seg_losses_all_ignored_is_nan():
    """F.cross_entropy with every pixel ignored returns NaN in the reference."""
    lib, st = _lib()
    N, K, H, W = 1, 5, 8, 8
    zw, zs, t, scb, mask = _loss_inputs(N, K, H, W, 5, all_ignored=True)
    sums = torch.zeros(6, dtype=torch.float64, device=dev())
    nws = lib.pp_seg_losses_workspace(N, H * W)
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    lib.pp_seg_losses_fwd(zw.to(dev()).data_ptr(), None, t.to(dev()).data_ptr(), None, N, K, H * W, K, 0, 0,
                          sums.data_ptr(), ws.data_ptr(), nws, st)
    lp = torch.zeros((), device=dev())
    lib.pp_losses_finalize(sums.data_ptr(), 0, lp.data_ptr(), None, None, st)
    assert math.isnan(float(lp))
    assert math.isnan(float(O.partial_cross_entropy_loss(zw, t, K)))


def test_aux_pce():
    lib, st = _lib()
    N, K, h, w, H, W = 2, 5, 8, 8, 64, 64
    g = torch.Generator().manual_seed(11)
    lo = torch.randn(N, K, h, w, generator=g)
    t = torch.randint(0, K + 1, (N, H, W), generator=g)
    t[torch.rand(N, H, W, generator=g) > 0.1] = K
    lor = lo.double().requires_grad_(True)
    up = F.interpolate(lor, size=(H, W), mode='bilinear', align_corners=True)
    loss = O.partial_cross_entropy_loss(up, t, K)
    (0.01 * loss).backward()
    lod, td = lo.to(dev()), t.to(dev())
    upd = torch.empty(N, K, H, W, device=dev())
    sums = torch.zeros(2, dtype=torch.float64, device=dev())
    ws = torch.empty(1 << 16, dtype=torch.uint8, device=dev())
    lib.pp_aux_pce_fwd(lod.data_ptr(), N, K, h, w, H, W, td.data_ptr(), K, upd.data_ptr(), sums.data_ptr(), ws.data_ptr(), 1 << 16, st)
    lv = torch.zeros((), device=dev())
    lib.pp_losses_finalize(sums.data_ptr(), 0, lv.data_ptr(), None, None, st)
    assert rel(upd, up) < 1e-5
    assert abs(float(lv) - float(loss)) < 1e-5 * abs(float(loss))
    dlo = torch.empty(N, K, h, w, device=dev())
    gg = torch.tensor(0.01, device=dev())
    lib.pp_aux_pce_bwd(upd.data_ptr(), td.data_ptr(), K, gg.data_ptr(), 1.0, sums.data_ptr(), dlo.data_ptr(), N, K, h, w, H, W, st)
    assert rel(dlo, lor.grad) < TOL


@pytest.mark.parametrize('mode,hid', [('cosine_similarity', 64), ('mean', 8), ('cosine_similarity', 200)])
def test_memory_update_and_ce(mode, hid):
    lib, st = _lib()
    K, h, w, H, W, B = 5, 8, 8, 64, 64, 2
    g = torch.Generator().manual_seed(hid)
    feat = torch.randn(B, hid, h, w, generator=g)
    t = torch.randint(0, K, (B, H, W), generator=g)
    t[torch.rand(B, H, W, generator=g) > 0.05] = K
    t[0][t[0] == 3] = K                                       # class 3 absent from sample 0
    scb = F.one_hot(t, K + 1).permute(0, 3, 1, 2).float().contiguous()
    args = O.default_args(hid_ch=hid, ensemble_mode=mode, epoch=400)
    bank_ref = torch.zeros(K, hid, 1, 1)
    bank_ref[1, :, 0, 0] = torch.randn(hid, generator=g)      # class 1 visited before, the others first visit
    bank_ref[3, :, 0, 0] = torch.randn(hid, generator=g)
    bank_d = bank_ref.clone().to(dev())
    O.memory_update(bank_ref, feat, scb, 37, args)
    featd = nhwc(feat).to(dev())
    mom = O.ramp_up_mo(37, 400, 0.9)
    ws = torch.empty(lib.pp_memory_update_workspace(K, hid) + 64, dtype=torch.uint8, device=dev())
    lib.pp_memory_update(featd.data_ptr(), hid, hid, h, w, scb.to(dev()).data_ptr(), K, H, W, bank_d.data_ptr(), mom,
                         1 if mode == 'cosine_similarity' else 0, ws.data_ptr(), ws.numel(), st)
    assert rel(bank_d, bank_ref) < 1e-5
    assert torch.equal(bank_d[3].cpu(), bank_ref[3]), 'class without scribble in sample 0 must be untouched'
    # bank classification CE and its gradient
    wfc = torch.randn(K, hid, 1, 1, generator=g)
    wr = wfc.double().requires_grad_(True)
    loss = O.cross_entropy_loss(F.conv2d(bank_ref.double(), wr).squeeze(-1).squeeze(-1), torch.arange(K))
    (1.5 * loss).backward()
    lv = torch.zeros((), device=dev())
    wd = wfc.to(dev())
    lib.pp_memory_ce_fwd(bank_d.data_ptr(), wd.data_ptr(), K, hid, lv.data_ptr(), st)
    assert abs(float(lv) - float(loss)) < 1e-5 * abs(float(loss))
    dw = torch.zeros(K, hid, device=dev())
    gg = torch.tensor(1.5, device=dev())
    lib.pp_memory_ce_bwd(bank_d.data_ptr(), wd.data_ptr(), K, hid, gg.data_ptr(), 1.0, dw.data_ptr(), 0, st)
    assert rel(dw, wr.grad.view(K, hid)) < TOL


def test_adam_matches_restatement():
    lib, st = _lib()
    g = torch.Generator().manual_seed(2)
    n = 10007
    p0 = torch.randn(n, generator=g)
    sd = {'w': p0.clone()}
    adam = O.AdamState()
    pd = torch.zeros(10008, device=dev()); pd[:n] = p0.to(dev())
    md, vd = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 10 ** (-step)
        adam.step(sd, {'w': gr}, 1e-4 * step, 3e-4)
        gd = torch.zeros(10008, device=dev()); gd[:n] = gr.to(dev())
        lib.pp_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 1e-4 * step, 0.9, 0.999, 1e-8, 3e-4, step, st)
        assert float((pd[:n].cpu() - sd['w']).abs().max()) < 2e-7
    assert float(pd[n:].abs().max()) == 0.0


def test_dice_counts():
    from pacingpseudo_amd.utils.metrics import batch_dice
    g = torch.Generator().manual_seed(4)
    N, K, H, W = 3, 5, 32, 32
    logits = torch.randn(N, K, H, W, generator=g)
    lab = torch.randint(0, K - 1, (N, H, W), generator=g)        # class K-1 never present in the label
    logits[:, K - 1] = -50.0                                      # ... nor predicted -> NaN entry
    onehot = F.one_hot(lab, K).permute(0, 3, 1, 2).float().contiguous()
    got = batch_dice(logits.to(dev()), onehot.to(dev()))
    sm = torch.softmax(logits, 1).numpy()
    ref = np.asarray([O.compute_dice(sm[n], onehot.numpy()[n]) for n in range(N)])
    assert np.allclose(got, ref, atol=1e-6, equal_nan=True)
    assert np.isnan(got[:, K - 1]).all()


F16X3_CASES = [
    # B, H, W, Cin, Cout, dil
    (2, 32, 32, 32, 32, 1),
    (1, 32, 32, 64, 64, 2),
    (2, 16, 16, 1024, 512, 1),      # deep K, 128 x 128 tiles
    (1, 16, 16, 192, 64, 1),
    (2, 8, 8, 12, 20, 1),           # ragged channels
    (3, 10, 6, 8, 4, 4),            # ragged pixel count, dilation 4
    (6, 128, 128, 32, 32, 1),       # halo-tile f16x3 kernel, two rows per wave, more tiles than blocks
    (5, 128, 64, 64, 32, 1),        # ... two channel chunks; data gradient with two output-channel groups
    (2, 12, 96, 96, 64, 1),         # ... one row per wave (three chunks, H % 8 != 0)
    # two-half halo kernel / two-pair weight gradient at awkward geometries: one tile only (second half all ghost
    # stages), odd tile counts, tiles_x = 3 and 7 (not powers of two), many N-blocks, more blocks than tiles
    (1, 4, 32, 32, 32, 1),
    (1, 4, 32, 64, 64, 1),
    (3, 20, 96, 64, 96, 1),
    (1, 36, 224, 32, 64, 1),
    (2, 8, 64, 64, 192, 1),
    (1, 44, 32, 32, 160, 1),
    (7, 4, 64, 64, 32, 1),
    # split-K halo launches (two 96-channel halves, the second accumulating): dec2.c1's shape class, several tiles per block,
    # and the accumulate-into-existing-output call on top of it
    (2, 32, 64, 192, 64, 1),
    (3, 12, 32, 192, 96, 1),
    # two-pair weight gradient over an ODD number of 32-channel chunks (dec1.c1: 96 -> 32; 160 -> 96): the last block group is
    # half empty -- its second pair stages nothing, multiplies nothing and writes nothing
    (2, 8, 64, 96, 32, 1),
    (1, 12, 32, 160, 96, 1),
    # the benchmark's 128-channel classes at tile-aligned geometries (implicit GEMM; round 5 ran them through a weight-streaming
    # two-half kernel as well: equal speed, removed), eight chunks, one tile only, odd tile counts, N = 256
    (2, 8, 64, 128, 128, 1),
    (1, 16, 32, 128, 256, 1),
    (2, 8, 32, 256, 128, 1),
    (1, 4, 32, 128, 32, 1),
    (3, 20, 96, 128, 64, 1),
    (1, 4, 32, 96, 32, 1),
]


@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', F16X3_CASES)
def test_conv3x3_f16x3(B, H, W, Cin, Cout, dil):
    """Split-fp16 ("f16x3") convolution against nn.Conv2d / its input gradient in fp64: same 1e-4 bar as the fp32
    kernels, plus the dynamic-range path (gradients of magnitude 1e-7 scaled through a device amax)."""
    lib, st = _lib()
    g = torch.Generator().manual_seed(B * 77 + Cin + Cout + dil)
    x = torch.randn(B, Cin, H, W, generator=g) * 3.0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g) * 1e-7          # far below the fp16 normal range
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double(), b.double(), 1, dil, dil)
    yr.backward(dy.double())
    ipad = (Cin + 3) // 4 * 4
    ld_in, ld_out = ipad + 8, (Cout + 7) // 4 * 4
    xin = torch.zeros(B, H, W, ld_in, device=dev()); xin[..., :Cin] = nhwc(x).to(dev())
    wf = torch.zeros(Cout, 9, ipad, device=dev()); wb = torch.zeros(Cin, 9, Cout, device=dev())
    lib.pp_pack_conv3x3_weights_f16x3(w.to(dev()).data_ptr(), Cout, Cin, ipad, wf.data_ptr(),
                                      wb.data_ptr() if (ipad == Cin and Cout % 4 == 0) else None, st)
    out = torch.full((B, H, W, ld_out), 7.0, device=dev())
    lib.pp_conv3x3_fwd_f16x3(xin.data_ptr(), ld_in, ipad, wf.data_ptr(), b.to(dev()).data_ptr(), out.data_ptr(), ld_out, Cout,
                             B, H, W, dil, 0, None, st)
    torch.cuda.synchronize()
    assert rel(nchw(out[..., :Cout]), yr) < TOL
    assert torch.all(out[..., Cout:] == 7.0)
    lib.pp_conv3x3_fwd_f16x3(xin.data_ptr(), ld_in, ipad, wf.data_ptr(), None, out.data_ptr(), ld_out, Cout, B, H, W, dil, 1,
                             None, st)
    assert rel(nchw(out[..., :Cout]), 2 * yr - b.double().view(1, -1, 1, 1)) < TOL
    if ipad == Cin and Cout % 4 == 0:
        dz = torch.zeros(B, H, W, ld_out, device=dev()); dz[..., :Cout] = nhwc(dy).to(dev())
        amax = dz.abs().max().reshape(1)
        dx = torch.full((B, H, W, ld_in), 3.0, device=dev())
        lib.pp_conv3x3_bwd_data_f16x3(dz.data_ptr(), ld_out, Cout, wb.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W, dil, 0,
                                      amax.data_ptr(), st)
        torch.cuda.synchronize()
        assert rel(nchw(dx[..., :Cin]), xr.grad) < TOL
        assert torch.all(dx[..., Cin:] == 3.0)
        # weight gradient (split-fp16 halo kernel where the shape qualifies, fp32 kernels otherwise), then accumulate
        wr = w.double().requires_grad_(True)
        F.conv2d(x.double(), wr, b.double(), 1, dil, dil).backward(dy.double())
        nws = lib.pp_conv3x3_bwd_weight_workspace(Cout, ipad, B, H, W)
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev())
        for acc in (0, 1):
            lib.pp_conv3x3_bwd_weight_f16x3(dz.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, ipad, Cin, B, H, W, dil,
                                            dw.data_ptr(), acc, ws.data_ptr(), nws, amax.data_ptr(), st)
            torch.cuda.synchronize()
            assert rel(dw, (acc + 1) * wr.grad) < TOL


WINO_CASES = [
    # B, H, W, Cin, Cout, dil
    (2, 16, 16, 128, 128, 1),
    (1, 16, 16, 256, 64, 2),        # dilation 2: four interleaved sub-images
    (2, 32, 32, 128, 256, 4),       # dilation 4 on a 32x32 map: sixteen 8x8 sub-images (encoder stage 6)
    (1, 8, 12, 12, 20, 1),          # ragged channel counts, non-square
    (3, 4, 4, 1024, 512, 1),        # deep K
    (1, 8, 8, 64, 192, 1),          # 192 outputs: three 64-row weight-gradient blocks
    (1, 8, 8, 40, 72, 1),           # pre-split GEMM: K tail (40 = 32 + 8), ragged M (4 tiles) and N (72) -- 256 x 64 tiles
    (1, 16, 16, 384, 128, 1),       # N = 128 forward / N = 384 data gradient: 256 x 128 tiles
]


@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', WINO_CASES)
def test_winograd_conv(B, H, W, Cin, Cout, dil):
    """Winograd F(2x2,3x3) path against nn.Conv2d / its autograd (fp64 reference)."""
    lib, st = _lib()
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout + dil)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), 1, dil, dil)
    yr.backward(dy.double())
    ld_in, ld_out = Cin + 8, Cout + 4
    xin = torch.zeros(B, H, W, ld_in, device=dev()); xin[..., :Cin] = nhwc(x).to(dev())
    dz = torch.zeros(B, H, W, ld_out, device=dev()); dz[..., :Cout] = nhwc(dy).to(dev())
    wd = w.to(dev())
    tile = lib.pp_conv3x3_wino_tile(H, W, dil)
    planes = (tile + 2) ** 2
    Uf = torch.empty(planes, Cout, Cin, device=dev()); Ub = torch.empty(planes, Cin, Cout, device=dev())
    lib.pp_wino_pack_weights(wd.data_ptr(), Cout, Cin, tile, Uf.data_ptr(), Ub.data_ptr(), st)
    nws = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, H, W, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, H, W, dil),
              lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, H, W, dil))
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
    out = torch.full((B, H, W, ld_out), 7.0, device=dev())
    lib.pp_conv3x3_wino_fwd(xin.data_ptr(), ld_in, Cin, Uf.data_ptr(), b.to(dev()).data_ptr(), out.data_ptr(), ld_out, Cout,
                            B, H, W, dil, 0, None, ws.data_ptr(), nws, st)
    assert rel(nchw(out[..., :Cout]), yr) < TOL
    assert torch.all(out[..., Cout:] == 7.0)
    vk = torch.empty(lib.pp_conv3x3_wino_vkeep_elems(Cin, B, H, W, dil), device=dev())
    lib.pp_conv3x3_wino_fwd(xin.data_ptr(), ld_in, Cin, Uf.data_ptr(), None, out.data_ptr(), ld_out, Cout, B, H, W, dil, 1,
                            vk.data_ptr(), ws.data_ptr(), nws, st)
    assert rel(nchw(out[..., :Cout]), 2 * yr - b.double().view(1, -1, 1, 1)) < TOL
    dx = torch.full((B, H, W, ld_in), 3.0, device=dev())
    lib.pp_conv3x3_wino_bwd_data(dz.data_ptr(), ld_out, Cout, Ub.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W, dil, 0,
                                 ws.data_ptr(), nws, st)
    assert rel(nchw(dx[..., :Cin]), xr.grad) < TOL
    assert torch.all(dx[..., Cin:] == 3.0)
    dw = torch.zeros(Cout, Cin, 3, 3, device=dev())
    lib.pp_conv3x3_wino_bwd_weight(dz.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, Cin, B, H, W, dil, dw.data_ptr(), 0,
                                   None, ws.data_ptr(), nws, st)
    assert rel(dw, wr.grad) < TOL
    lib.pp_conv3x3_wino_bwd_weight(dz.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, Cin, B, H, W, dil, dw.data_ptr(), 1,
                                   vk.data_ptr(), ws.data_ptr(), nws, st)       # transformed input kept by the forward call
    assert rel(dw, 2 * wr.grad) < TOL
    if tile == 4 and Cin % 8 == 0 and Cout % 8 == 0:
        # split-fp16 GEMM in the Winograd domain on pre-split operands: same bar; the data gradient runs with 1e-7-sized inputs
        # (operand scaling from max |dz|: once found by the library, once brought by the caller as the engine does)
        Uf16 = torch.empty_like(Uf); Ub16 = torch.empty_like(Ub)
        lib.pp_wino_pack_weights_f16x3(wd.data_ptr(), Cout, Cin, tile, Uf16.data_ptr(), Ub16.data_ptr(), st)
        out.fill_(7.0)
        lib.pp_conv3x3_wino_fwd_f16x3(xin.data_ptr(), ld_in, Cin, Uf16.data_ptr(), b.to(dev()).data_ptr(), out.data_ptr(), ld_out,
                                      Cout, B, H, W, dil, 0, vk.data_ptr(), ws.data_ptr(), nws, st)
        assert rel(nchw(out[..., :Cout]), yr) < TOL
        assert torch.all(out[..., Cout:] == 7.0)
        dx.fill_(3.0)
        dz_small = dz * 1e-7
        lib.pp_conv3x3_wino_bwd_data_f16x3(dz_small.data_ptr(), ld_out, Cout, Ub16.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W,
                                           dil, 0, ws.data_ptr(), nws, None, st)
        assert rel(nchw(dx[..., :Cin]) * 1e7, xr.grad) < TOL
        assert torch.all(dx[..., Cin:] == 3.0)
        amax = dz_small.abs().max().reshape(1).contiguous()
        dx.fill_(3.0)
        lib.pp_conv3x3_wino_bwd_data_f16x3(dz_small.data_ptr(), ld_out, Cout, Ub16.data_ptr(), dx.data_ptr(), ld_in, Cin, B, H, W,
                                           dil, 0, ws.data_ptr(), nws, amax.data_ptr(), st)
        assert rel(nchw(dx[..., :Cin]) * 1e7, xr.grad) < TOL
        assert torch.all(dx[..., Cin:] == 3.0)
        # weight gradient: own input transform, then the transformed input kept by the f16x3 forward call above
        dw.zero_()
        lib.pp_conv3x3_wino_bwd_weight_f16x3(dz_small.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, Cin, B, H, W, dil,
                                             dw.data_ptr(), 0, None, ws.data_ptr(), nws, None, st)
        assert rel(dw * 1e7, wr.grad) < TOL
        lib.pp_conv3x3_wino_bwd_weight_f16x3(dz_small.data_ptr(), ld_out, Cout, xin.data_ptr(), ld_in, Cin, B, H, W, dil,
                                             dw.data_ptr(), 1, vk.data_ptr(), ws.data_ptr(), nws, amax.data_ptr(), st)
        assert rel(dw * 1e7, 2 * wr.grad) < TOL


# ---- --is_stride_conv / --is_trans_conv (models/unet.py:100-152) ---------------------------------------------------------
@pytest.mark.parametrize('Cin,Cout,k,N,H,W', [(32, 16, 2, 2, 8, 8), (32, 32, 1, 2, 8, 8), (128, 64, 2, 2, 16, 16), (20, 12, 2, 1, 5, 7),
                                              (8, 4, 2, 2, 70, 66), (512, 256, 2, 2, 14, 14)])
def test_convtranspose(Cin, Cout, k, N, H, W):
    """nn.ConvTranspose2d(Cin, Cout, k, k, bias=False) forward, data gradient and weight gradient (split reduction when the
    pixel count exceeds one split) against aten in fp64; the up-sampled tensor is written into a wider concat buffer."""
    lib, st = _lib()
    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) / math.sqrt(Cin)
    dy = torch.randn(N, Cout, k * H, k * W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, k)
    yr.backward(dy.double())
    xd, wd = nhwc(x).to(dev()), w.to(dev())
    ld = Cout + 8                                                   # the [up | skip] concat buffer of the decoder
    cat = torch.full((N, k * H, k * W, ld), 7.0, device=dev())
    lib.pp_convtranspose_fwd(xd.data_ptr(), Cin, Cin, wd.data_ptr(), cat.data_ptr(), ld, Cout, k, N, H, W, st)
    assert rel(nchw(cat[..., :Cout]), yr) < 1e-5
    assert bool((cat[..., Cout:] == 7.0).all()), 'wrote outside its channel slice'
    gcat = torch.zeros(N, k * H, k * W, ld, device=dev())
    gcat[..., :Cout] = nhwc(dy).to(dev())
    dxd = torch.empty(N, H, W, Cin, device=dev())
    lib.pp_convtranspose_bwd_data(gcat.data_ptr(), ld, Cout, wd.data_ptr(), dxd.data_ptr(), Cin, Cin, k, N, H, W, 0, st)
    assert rel(nchw(dxd), xr.grad) < 1e-5
    lib.pp_convtranspose_bwd_data(gcat.data_ptr(), ld, Cout, wd.data_ptr(), dxd.data_ptr(), Cin, Cin, k, N, H, W, 1, st)
    assert rel(nchw(dxd), 2 * xr.grad) < 1e-5
    nb = lib.pp_convtranspose_bwd_weight_workspace(Cin, Cout, k, N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    dwd = torch.empty_like(wd)
    lib.pp_convtranspose_bwd_weight(gcat.data_ptr(), ld, Cout, xd.data_ptr(), Cin, Cin, k, N, H, W, dwd.data_ptr(), 0,
                                    ws.data_ptr(), nb, st)
    assert rel(dwd, wr.grad) < 1e-5
    with pytest.raises(RuntimeError):
        lib.pp_convtranspose_bwd_weight(gcat.data_ptr(), ld, Cout, xd.data_ptr(), Cin, Cin, k, N, H, W, dwd.data_ptr(), 0,
                                        ws.data_ptr(), nb - 512, st)
    with pytest.raises(RuntimeError):
        lib.pp_convtranspose_fwd(xd.data_ptr(), Cin, Cin, wd.data_ptr(), cat.data_ptr(), ld, Cout, 3, N, H, W, st)


@pytest.mark.parametrize('C,N,Ho,Wo', [(32, 2, 8, 8), (4, 1, 3, 5), (64, 2, 56, 56)])
def test_stride2_gather_scatter(C, N, Ho, Wo):
    """z[:, y, x] = full[:, 2y, 2x]; the scatter is its exact adjoint (zeros elsewhere).  Bit-exact: both only move data."""
    lib, st = _lib()
    full = torch.randn(N, 2 * Ho, 2 * Wo, C, device=dev())
    z = torch.empty(N, Ho, Wo, C, device=dev())
    lib.pp_stride2_gather(full.data_ptr(), C, z.data_ptr(), C, C, N, Ho, Wo, st)
    assert torch.equal(z, full[:, ::2, ::2].contiguous())
    back = torch.full((N, 2 * Ho, 2 * Wo, C), 3.0, device=dev())
    lib.pp_stride2_scatter(z.data_ptr(), C, back.data_ptr(), C, C, N, Ho, Wo, st)
    want = torch.zeros_like(full)
    want[:, ::2, ::2] = z
    assert torch.equal(back, want)


@pytest.mark.parametrize('Cin,Cout,N,H,W', [(4, 8, 2, 16, 16), (32, 64, 2, 32, 32), (256, 512, 2, 8, 8)])
def test_stride2_convolution_is_sampled_stride1(Cin, Cout, N, H, W):
    """The identity the engine builds --is_stride_conv on: conv(x, w, stride 2, pad 1) and its two gradients from the stride-1
    kernels + gather / scatter, against aten's strided convolution in fp64."""
    lib, st = _lib()
    g = torch.Generator().manual_seed(Cin * 3 + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g)
    dz = torch.randn(N, Cout, H // 2, W // 2, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    zr = F.conv2d(xr, wr, b.double(), 2, 1)
    zr.backward(dz.double())
    xd = nhwc(x).to(dev())
    wf, wb = pack_w(w.to(dev()), Cin)
    zf = torch.empty(N, H, W, Cout, device=dev())
    lib.pp_conv3x3_fwd(xd.data_ptr(), Cin, Cin, wf.data_ptr(), b.to(dev()).data_ptr(), zf.data_ptr(), Cout, Cout, N, H, W, 1, 0, st)
    z = torch.empty(N, H // 2, W // 2, Cout, device=dev())
    lib.pp_stride2_gather(zf.data_ptr(), Cout, z.data_ptr(), Cout, Cout, N, H // 2, W // 2, st)
    assert rel(nchw(z), zr) < TOL
    dzf = torch.empty(N, H, W, Cout, device=dev())
    lib.pp_stride2_scatter(nhwc(dz).to(dev()).data_ptr(), Cout, dzf.data_ptr(), Cout, Cout, N, H // 2, W // 2, st)
    dx = torch.empty(N, H, W, Cin, device=dev())
    lib.pp_conv3x3_bwd_data(dzf.data_ptr(), Cout, Cout, wb.data_ptr(), dx.data_ptr(), Cin, Cin, N, H, W, 1, 0, st)
    assert rel(nchw(dx), xr.grad) < TOL
    nb = lib.pp_conv3x3_bwd_weight_workspace(Cout, Cin, N, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev())
    dw = torch.empty(Cout, Cin, 3, 3, device=dev())
    lib.pp_conv3x3_bwd_weight(dzf.data_ptr(), Cout, Cout, xd.data_ptr(), Cin, Cin, Cin, N, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nb, st)
    assert rel(dw, wr.grad) < TOL


_PSP_SCRIPT = r'''
import math, sys, torch
sys.path.insert(0, sys.argv[1])
from pacingpseudo_amd._lib import lib, stream_ptr
B, H, W, Cin, Cout, dil = (int(v) for v in sys.argv[3:9])
dev = torch.device('cuda', 0); st = stream_ptr()
g = torch.Generator().manual_seed(B + Cin + Cout)
x = torch.randn(B, H, W, Cin, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev)
b = torch.randn(Cout, generator=g).to(dev)
dz = (torch.randn(B, H, W, Cout, generator=g) * 1e-3).to(dev)
Uf = torch.empty(36, Cout, Cin, device=dev); Ub = torch.empty(36, Cin, Cout, device=dev)
lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, 4, Uf.data_ptr(), Ub.data_ptr(), st)
nws = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, H, W, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, H, W, dil))
ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
out = torch.empty(B, H, W, Cout, device=dev)
lib.pp_conv3x3_wino_fwd_f16x3(x.data_ptr(), Cin, Cin, Uf.data_ptr(), b.data_ptr(), out.data_ptr(), Cout, Cout, B, H, W, dil, 0, None,
                              ws.data_ptr(), nws, st)
dx = torch.empty(B, H, W, Cin, device=dev)
lib.pp_conv3x3_wino_bwd_data_f16x3(dz.data_ptr(), Cout, Cout, Ub.data_ptr(), dx.data_ptr(), Cin, Cin, B, H, W, dil, 0, ws.data_ptr(), nws,
                                   None, st)
torch.cuda.synchronize()
torch.save(dict(x=x.cpu(), w=w.cpu(), b=b.cpu(), dz=dz.cpu(), out=out.cpu(), dx=dx.cpu()), sys.argv[2])
'''


@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', [(32, 32, 32, 256, 256, 1), (16, 32, 32, 512, 128, 2)])
def test_winograd_gemm_persistent_form(tmp_path, B, H, W, Cin, Cout, dil):
    """wino_gemm_psp_kernel (round 5: persistent over its tiles, one LDS-DMA pipeline across tile boundaries, vmcnt(63) window behind the
    accumulator stores) -- launched for shapes with >= 512 full tiles, which the small per-op cases above never reach: forward and
    data gradient of the split-fp16 F(4x4,3x3) convolution against nn.Conv2d in fp64, and BIT-identical to the one-tile-per-block
    kernel of round 3 (PP_WINO_GEMM_PERSIST=0 in a second process: the K order inside a tile is the same)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ('1', '0'):
        path = str(tmp_path / f'psp{mode}.pt')
        r = subprocess.run([sys.executable, '-c', _PSP_SCRIPT, root, path] + [str(v) for v in (B, H, W, Cin, Cout, dil)],
                           env=dict(os.environ, PP_WINO_GEMM_PERSIST=mode), capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = torch.load(path)
    p, o = res['1'], res['0']
    assert torch.equal(p['out'], o['out']) and torch.equal(p['dx'], o['dx'])
    xr = p['x'].permute(0, 3, 1, 2).double().requires_grad_(True)
    yr = F.conv2d(xr, p['w'].double(), p['b'].double(), 1, dil, dil)
    yr.backward(p['dz'].permute(0, 3, 1, 2).double())
    assert rel(p['out'].permute(0, 3, 1, 2), yr) < TOL
    assert rel(p['dx'].permute(0, 3, 1, 2), xr.grad) < TOL
