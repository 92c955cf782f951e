"""Guard bands: every output tensor and every workspace of the hot kernels is embedded in a larger
allocation whose surroundings hold a sentinel -- PAST THE LAST PIXEL and IN FRONT of the first one, and past the stated
workspace size; after the launch the sentinels must be intact (a write one element out of bounds is caught here, where the
channel-slice canaries of test_gpu_ops.py only see writes into the padding columns of a row)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SENT = 1234.5
GUARD = 8192          # floats on either side


def _lib():
    from pacingpseudo_amd._lib import lib, stream_ptr
    return lib, stream_ptr()


class Guarded:
    """`numel` floats with a sentinel band on both sides; `.t` is the inner tensor (16-byte aligned)."""

    def __init__(self, numel, fill=0.0):
        self.full = torch.full((numel + 2 * GUARD,), SENT, device='cuda')
        self.t = self.full[GUARD:GUARD + numel]
        self.t.fill_(fill)

    def ok(self):
        return bool((self.full[:GUARD] == SENT).all()) and bool((self.full[GUARD + self.t.numel():] == SENT).all())


class GuardedWs:
    """Workspace of exactly `nbytes` with a sentinel band behind it."""

    def __init__(self, nbytes):
        pad = (-nbytes) % 16
        self.n = nbytes
        self.full = torch.full((nbytes + pad + 4 * GUARD,), 0xA5, dtype=torch.uint8, device='cuda')

    def ptr(self):
        return self.full.data_ptr()

    def ok(self):
        return bool((self.full[self.n + ((-self.n) % 16):] == 0xA5).all())


CONV_CASES = [   # B, H, W, Cin, Cout, dil -- the kernel each shape selects at the benchmark geometry
    (2, 64, 64, 64, 64, 1),       # two-half halo kernel (narrow high-resolution layers)
    (2, 32, 32, 96, 32, 1),       # one-half halo kernel (Cin = 96)
    (1, 32, 32, 128, 128, 1),     # split-fp16 implicit GEMM 128 x 128
    (1, 32, 32, 192, 64, 1),      # ... 128 x 64 tiles
    (2, 16, 16, 256, 256, 1),     # Winograd F(4x4), pre-split GEMM 128 x 256
    (1, 16, 16, 512, 128, 2),     # Winograd, dilation 2, 256 x 128 tiles
    (1, 12, 20, 64, 40, 1),       # ragged: fallback kernels
]


@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', CONV_CASES)
def test_convolution_family_stays_inside_its_buffers(B, H, W, Cin, Cout, dil):
    lib, st = _lib()
    g = torch.Generator().manual_seed(Cin + Cout)
    P = B * H * W
    x = torch.randn(P * Cin, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    dzs = (torch.randn(P * Cout, generator=g) * 1e-3).cuda()
    amax = dzs.abs().max().reshape(1).contiguous()
    wino = Cin >= 256 and H % (4 * dil) == 0 and W % (4 * dil) == 0
    out, dx, dw = Guarded(P * Cout), Guarded(P * Cin), Guarded(Cout * Cin * 9)
    if wino:
        Uf, Ub = torch.empty(36, Cout, Cin, device='cuda'), torch.empty(36, Cin, Cout, device='cuda')
        lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, 4, Uf.data_ptr(), Ub.data_ptr(), st)
        nws = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, H, W, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, H, W, dil),
                  lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, H, W, dil))
        ws = GuardedWs(nws)
        vk = Guarded(lib.pp_conv3x3_wino_vkeep_elems(Cin, B, H, W, dil))
        lib.pp_conv3x3_wino_fwd_f16x3(x.data_ptr(), Cin, Cin, Uf.data_ptr(), bias.data_ptr(), out.t.data_ptr(), Cout, Cout, B, H, W,
                                      dil, 0, vk.t.data_ptr(), ws.ptr(), nws, st)
        lib.pp_conv3x3_wino_bwd_data_f16x3(dzs.data_ptr(), Cout, Cout, Ub.data_ptr(), dx.t.data_ptr(), Cin, Cin, B, H, W, dil, 0,
                                           ws.ptr(), nws, amax.data_ptr(), st)
        lib.pp_conv3x3_wino_bwd_weight_f16x3(dzs.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, B, H, W, dil, dw.t.data_ptr(), 0,
                                             vk.t.data_ptr(), ws.ptr(), nws, amax.data_ptr(), st)
        torch.cuda.synchronize()
        assert vk.ok(), 'kept transformed input'
    else:
        wf, wb = torch.zeros(Cout, 9, Cin, device='cuda'), torch.zeros(Cin, 9, Cout, device='cuda')
        lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), wb.data_ptr(), st)
        nws = lib.pp_conv3x3_bwd_weight_workspace(Cout, Cin, B, H, W)
        ws = GuardedWs(nws)
        lib.pp_conv3x3_fwd_f16x3(x.data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), out.t.data_ptr(), Cout, Cout, B, H, W, dil,
                                 0, None, st)
        lib.pp_conv3x3_bwd_data_f16x3(dzs.data_ptr(), Cout, Cout, wb.data_ptr(), dx.t.data_ptr(), Cin, Cin, B, H, W, dil, 0,
                                      amax.data_ptr(), st)
        lib.pp_conv3x3_bwd_weight_f16x3(dzs.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, Cin, B, H, W, dil, dw.t.data_ptr(), 0,
                                        ws.ptr(), nws, amax.data_ptr(), st)
        torch.cuda.synchronize()
    assert out.ok(), 'forward output'
    assert dx.ok(), 'data gradient'
    assert dw.ok(), 'weight gradient'
    assert ws.ok(), 'workspace'
    assert bool(torch.isfinite(out.t).all()) and float(out.t.abs().max()) > 0 and float(dw.t.abs().max()) > 0


@pytest.mark.parametrize('C,N,H,W', [(32, 2, 64, 64), (64, 3, 16, 24), (512, 2, 8, 8)])
def test_norm_and_spatial_kernels_stay_inside_their_buffers(C, N, H, W):
    lib, st = _lib()
    g = torch.Generator().manual_seed(C)
    P = N * H * W
    z = torch.randn(P * C, generator=g).cuda()
    dy = torch.randn(P * C, generator=g).cuda()
    scale, shift = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    mean, invstd = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    gamma = torch.ones(C, device='cuda')
    y, dz = Guarded(P * C), Guarded(P * C)
    lib.pp_bn_lrelu_fwd(z.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), y.t.data_ptr(), C, C, P, 1, 0.01, st)
    nws = lib.pp_bn_workspace(C, P, 1) + 12 * C
    ws = GuardedWs(nws)
    dgm, dbt, dbc = Guarded(C), Guarded(C), Guarded(C)
    am = Guarded(4)
    lib.pp_bn_lrelu_bwd_amax(dy.data_ptr(), C, z.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                             gamma.data_ptr(), 1, dz.t.data_ptr(), C, dgm.t.data_ptr(), dbt.t.data_ptr(), dbc.t.data_ptr(), 0, C, P, 1,
                             0.01, ws.ptr(), nws, am.t.data_ptr(), st)
    pooled, dpool = Guarded(P * C // 4), Guarded(P * C, fill=1.0)
    lib.pp_maxpool2_fwd(y.t.data_ptr(), C, pooled.t.data_ptr(), C, C, N, H, W, st)
    lib.pp_maxpool2_bwd(y.t.data_ptr(), C, pooled.t.data_ptr(), C, dpool.t.data_ptr(), C, C, N, H, W, 1, st)
    up, dlow = Guarded(P * C * 4), Guarded(P * C)
    lib.pp_bilinear_fwd(y.t.data_ptr(), C, up.t.data_ptr(), C, C, N, H, W, 2 * H, 2 * W, st)
    lib.pp_bilinear_bwd(up.t.data_ptr(), C, dlow.t.data_ptr(), C, C, N, H, W, 2 * H, 2 * W, 0, st)
    torch.cuda.synchronize()
    for name, gbuf in dict(y=y, dz=dz, dgamma=dgm, dbeta=dbt, dbias=dbc, amax=am, pooled=pooled, dpool=dpool, up=up, dlow=dlow).items():
        assert gbuf.ok(), name
    assert ws.ok(), 'BatchNorm workspace'


def test_loss_and_optimizer_kernels_stay_inside_their_buffers():
    lib, st = _lib()
    g = torch.Generator().manual_seed(0)
    B, K, H, W = 3, 5, 24, 40
    logits = torch.randn(2 * B, K, H, W, generator=g).cuda()
    scb = torch.nn.functional.one_hot(torch.randint(0, K + 1, (B, H, W), generator=g), K + 1).permute(0, 3, 1, 2).float().contiguous().cuda()
    mask = torch.ones(B, 1, H, W, device='cuda')
    target = torch.empty(B, H, W, dtype=torch.int64, device='cuda')
    lib.pp_argmax_channels(scb.data_ptr(), B, K + 1, H * W, target.data_ptr(), st)
    nws = max(lib.pp_seg_losses_workspace(B, H * W), 1024 * 16)
    ws = GuardedWs(nws)
    sums = torch.zeros(8, dtype=torch.float64, device='cuda')
    lib.pp_seg_losses_fwd(logits.data_ptr(), logits[B:].data_ptr(), target.data_ptr(), mask.data_ptr(), B, K, H * W, K, 1, 1,
                          sums.data_ptr(), ws.ptr(), nws, st)
    dl = Guarded(2 * B * K * H * W)
    one = torch.ones((), device='cuda')
    lib.pp_seg_losses_bwd(logits.data_ptr(), logits[B:].data_ptr(), target.data_ptr(), mask.data_ptr(), B, K, H * W, K, 1, 1, 0,
                          sums.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), 1.0, dl.t.data_ptr(),
                          dl.t[B * K * H * W:].data_ptr(), st)
    n = 100003 * 4
    p, m, v = Guarded(n, 0.5), Guarded(n), Guarded(n)
    grad = torch.randn(n, generator=g).cuda()
    lib.pp_adam_step(p.t.data_ptr(), grad.data_ptr(), m.t.data_ptr(), v.t.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 3e-4, 1, st)
    torch.cuda.synchronize()
    assert dl.ok() and ws.ok() and p.ok() and m.ok() and v.ok()
    assert bool(torch.isfinite(dl.t).all()) and float((p.t - 0.5).abs().max()) > 0
