"""16-bit STORAGE modes (BASELINE config 5; include/pacingpseudo_hip_h16.h and, round 6, include/pacingpseudo_hip_bf16.h): every
`_h16` / `_bf16` entry point against its fp32 twin.  Every test of this module runs once per storage kind (fixture `storage_kind`):
IEEE fp16 (10 stored significand bits) and bfloat16 (7; the type BASELINE.json configs[4] names).  "fp16" / "2^-11" in the
comments below read "the 16-bit type" / "half an ulp of it" for the second kind.

The `_h16` kernels are the fp32 kernels compiled a second time with fp16 loads / stores (pacingpseudo_amd/csrc/pp_common.h,
PP_ACT_H16): arithmetic, accumulators and statistics stay fp32.  So on operands that ARE fp16 numbers the two forms must agree
up to (a) the rounding of an output tensor to fp16 -- half a unit in the last place, 2^-11 relative -- and (b) fp32 summation
order.  Every test below feeds both forms the same fp16-representable values and checks exactly that:

    activation outputs:   |h16 - fp32| <= 0.75 ulp16(fp32) + ATOL_SUM * max|fp32|     (rne16 of the fp32 result, up to ties)
    fp32 / fp64 outputs:  relative max-norm error <= TOL_F32 (weight gradients, statistics, logits)

The whole-step test at the end runs one training step of the benchmark network (full widths, 256 x 256) in both storage modes
and states the tolerance of the mode as a whole."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL_SUM = 4e-6       # fp32 summation-order noise of a reduction, relative to the largest output
TOL_F32 = 2e-5        # fp32 outputs of reductions over fp16-exact operands
SLOPE = 1e-2


class KIND:
    """The storage kind of the running test (set by the autouse fixture below)."""
    name, dtype, mant, emin, suffix = 'fp16', torch.float16, 10, -14, '_h16'
    half_ulp = 2.0 ** -11          # relative rounding error of one store in the 16-bit type


@pytest.fixture(autouse=True, params=['fp16', 'bf16'])
def storage_kind(request):
    if request.param == 'bf16':
        KIND.name, KIND.dtype, KIND.mant, KIND.emin, KIND.suffix = 'bf16', torch.bfloat16, 7, -126, '_bf16'
    else:
        KIND.name, KIND.dtype, KIND.mant, KIND.emin, KIND.suffix = 'fp16', torch.float16, 10, -14, '_h16'
    KIND.half_ulp = 2.0 ** -(KIND.mant + 1)
    yield request.param
    KIND.name, KIND.dtype, KIND.mant, KIND.emin, KIND.suffix, KIND.half_ulp = 'fp16', torch.float16, 10, -14, '_h16', 2.0 ** -11


def dev():
    return torch.device('cuda', 0)


def _libs():
    from pacingpseudo_amd._lib import lib, lib_for, stream_ptr
    return lib, lib_for(KIND.name), stream_ptr()


def r16(t):
    """Round to the nearest number of the 16-bit type, keep fp32."""
    return t.to(KIND.dtype).float()


def ulp16(t):
    """Unit in the last place of the 16-bit type at |t| (subnormal spacing below its smallest normal number)."""
    a = t.abs().double().clamp_min(2.0 ** KIND.emin)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - KIND.mant)


def check_act(h, f, what, intermediate=0.0):
    """h: fp16 tensor produced by the _h16 form, f: fp32 tensor produced by the fp32 form on the same operands.
    intermediate: largest magnitude of a tensor that this call stores in fp16 BEFORE the output is formed (the first half of a
    split-K sum, the z of an unfused BatchNorm epilogue): its rounding -- half an ulp at ITS magnitude -- reaches the output,
    which may be much smaller (cancellation), so the bound gains 2^-11 of that magnitude."""
    assert h.dtype == KIND.dtype and f.dtype == torch.float32, what
    f64, h64 = f.double(), h.double()
    assert bool(torch.isfinite(h64).all()), what
    m = float(f64.abs().max())
    err = (h64 - f64).abs()
    bound = 0.75 * ulp16(f64) + ATOL_SUM * max(m, 1e-30) + KIND.half_ulp * intermediate
    bad = err > bound
    assert not bool(bad.any()), (what, float(err.max()), float((err / bound).max()), int(bad.sum()))


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


class Pair:
    """The same buffers for both forms: `act` tensors are fp32 on one side and fp16 on the other, everything else identical."""

    def __init__(self):
        self.lib, self.lib16, self.st = _libs()
        self.keep = []

    def acts(self, t):
        """(fp32 device tensor, fp16 device tensor) holding the same fp16-representable values."""
        f = r16(t).to(dev()).contiguous()
        h = f.to(KIND.dtype).contiguous()
        self.keep += [f, h]
        return f, h

    def bufs(self, t):
        a, b = t.clone().to(dev()).contiguous(), t.clone().to(dev()).contiguous()
        self.keep += [a, b]
        return a, b

    def call(self, name, args32, args16):
        getattr(self.lib, name)(*args32, self.st)
        getattr(self.lib16, name)(*args16, self.st)
        torch.cuda.synchronize()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------ 3x3 convolution, direct kernels
@pytest.mark.parametrize('B,H,W,Cin,Cout,dil,mode', [
    (2, 8, 64, 32, 32, 1, 1),       # two-half halo kernel, train-mode statistics
    (2, 8, 32, 64, 64, 1, 2),       # two-half halo kernel, eval-mode epilogue
    (2, 8, 32, 96, 32, 1, 1),       # one-half halo kernel (three channel chunks)
    (2, 8, 32, 192, 64, 1, 1),      # split-K over two launches (second accumulates)
    (2, 16, 16, 128, 128, 1, 1),    # implicit GEMM 128 x 128
    (2, 12, 24, 64, 64, 1, 2),      # width not a multiple of 32: implicit GEMM 128 x 64
    (1, 16, 16, 64, 32, 2, 1),      # dilated: implicit GEMM 128 x 32
])
def test_conv3x3_forward_with_batchnorm_epilogue(B, H, W, Cin, Cout, dil, mode):
    """pp_conv3x3_fwd_bn_h16 (split-fp16 kernels, the activation operand has no low part) against pp_conv3x3_fwd_bn."""
    P = Pair()
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x32, x16 = P.acts(torch.randn(B, H, W, Cin, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    wf = torch.empty(Cout, 9, Cin, device=dev())
    P.lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), None, P.st)
    groups = 2 if B % 2 == 0 else 1
    scale, shift = (torch.rand(Cout, generator=g) + 0.5).to(dev()), torch.randn(Cout, generator=g).to(dev())
    nst = P.lib.pp_conv3x3_bn_stats_bytes(Cout, B, H, W, groups)
    outs = []
    for K, x, dt in ((P.lib, x32, torch.float32), (P.lib16, x16, KIND.dtype)):
        out = torch.zeros(B, H, W, Cout, device=dev(), dtype=dt)
        stats = torch.zeros(nst // 8 + 2, device=dev(), dtype=torch.float64)
        rows = ctypes.c_int(0)
        K.pp_conv3x3_fwd_bn(x.data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, H, W, dil, 1, None,
                            mode, scale.data_ptr(), shift.data_ptr(), SLOPE, groups, stats.data_ptr(), nst, ctypes.byref(rows), P.st)
        torch.cuda.synchronize()
        tot = None
        if mode == 1:
            tot = stats[:groups * rows.value * 2 * Cout].view(groups, rows.value, 2, Cout).sum(1)
        outs.append((out, tot))
    (o32, s32), (o16, s16) = outs
    # two calls keep an fp16 intermediate: the split-K pair (first half of the sum) and the shapes whose epilogue is not fused
    # (z is stored, then normalised in a second pass: W = 24 leaves no 128-pixel tiles per group)
    inter = 0.0
    if Cin == 192:
        inter = float(o32.abs().max())
    if mode == 2 and W == 24:
        inter = float(o32.abs().max()) * 2.0
    check_act(o16, o32, 'z / y', inter)
    if mode == 1:          # the statistics come from the fp32 accumulators, not from the rounded tensor
        assert rel(s16, s32) < (max(2e-4, 0.5 * KIND.half_ulp) if Cin == 192 else TOL_F32)      # (split-K: the first half-sum was stored in 16 bits)


def test_first_layer_kernels():
    """The 4-channel first layer: pp_conv3x3_fwd_bn_h16 (f16x3 = 0 -> the fp32-MFMA first-layer kernel with fp16 loads) and its
    weight gradient (pp_conv3x3_bwd_weight_h16)."""
    P = Pair()
    g = torch.Generator().manual_seed(5)
    B, H, W, Cout = 2, 16, 32, 32
    x = torch.zeros(B, H, W, 4)
    x[..., 0] = torch.randn(B, H, W, generator=g)
    x32, x16 = P.acts(x)
    w = (torch.randn(Cout, 1, 3, 3, generator=g) / 3).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    wf = torch.empty(Cout, 9, 4, device=dev())
    P.lib.pp_pack_conv3x3_weights(w.data_ptr(), Cout, 1, 4, wf.data_ptr(), None, P.st)
    nst = P.lib.pp_conv3x3_bn_stats_bytes(Cout, B, H, W, 2)
    res = []
    for K, xx, dt in ((P.lib, x32, torch.float32), (P.lib16, x16, KIND.dtype)):
        out = torch.zeros(B, H, W, Cout, device=dev(), dtype=dt)
        stats = torch.zeros(nst // 8 + 2, device=dev(), dtype=torch.float64)
        rows = ctypes.c_int(0)
        K.pp_conv3x3_fwd_bn(xx.data_ptr(), 4, 4, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, H, W, 1, 0, None, 1,
                            None, None, SLOPE, 2, stats.data_ptr(), nst, ctypes.byref(rows), P.st)
        torch.cuda.synchronize()
        res.append((out, stats[:2 * rows.value * 2 * Cout].view(2, rows.value, 2, Cout).sum(1)))
    check_act(res[1][0], res[0][0], 'first layer z')
    assert rel(res[1][1], res[0][1]) < TOL_F32
    dz32, dz16 = P.acts(torch.randn(B, H, W, Cout, generator=g) * 1e-2)
    nws = P.lib.pp_conv3x3_bwd_weight_workspace(Cout, 4, B, H, W)
    dws = []
    for K, d, xx in ((P.lib, dz32, x32), (P.lib16, dz16, x16)):
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
        dw = torch.zeros(Cout, 1, 3, 3, device=dev())
        K.pp_conv3x3_bwd_weight(d.data_ptr(), Cout, Cout, xx.data_ptr(), 4, 4, 1, B, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nws, P.st)
        torch.cuda.synchronize()
        dws.append(dw)
    assert rel(dws[1], dws[0]) < TOL_F32


@pytest.mark.parametrize('B,H,W,O,I,accumulate', [(2, 8, 32, 64, 32, 0), (2, 8, 64, 32, 64, 1), (2, 16, 16, 128, 128, 0), (2, 8, 32, 64, 192, 1)])
def test_conv3x3_data_gradient(B, H, W, O, I, accumulate):
    """pp_conv3x3_bwd_data_f16x3_h16: dz scaled by its amax (a power of two: an fp16 dz stays exact), read-add-write accumulate."""
    P = Pair()
    g = torch.Generator().manual_seed(O + I)
    dz32, dz16 = P.acts(torch.randn(B, H, W, O, generator=g) * 3e-3)
    w = (torch.randn(O, I, 3, 3, generator=g) / math.sqrt(9 * I)).to(dev())
    wb = torch.empty(I, 9, O, device=dev())
    wf = torch.empty(O, 9, I, device=dev())
    P.lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), O, I, I, wf.data_ptr(), wb.data_ptr(), P.st)
    amax = dz32.abs().max().reshape(1).contiguous()
    dx32, dx16 = P.acts(torch.randn(B, H, W, I, generator=g) * 1e-2)
    P.call('pp_conv3x3_bwd_data_f16x3',
           (dz32.data_ptr(), O, O, wb.data_ptr(), dx32.data_ptr(), I, I, B, H, W, 1, accumulate, amax.data_ptr()),
           (dz16.data_ptr(), O, O, wb.data_ptr(), dx16.data_ptr(), I, I, B, H, W, 1, accumulate, amax.data_ptr()))
    check_act(dx16, dx32, 'dx')


@pytest.mark.parametrize('B,H,W,O,C', [(2, 8, 32, 32, 32), (2, 8, 64, 64, 64), (2, 8, 32, 32, 64), (2, 8, 32, 128, 96),
                                        (2, 12, 24, 64, 64), (1, 14, 14, 256, 128)])
def test_conv3x3_weight_gradient(B, H, W, O, C):
    """pp_conv3x3_bwd_weight_f16x3_h16: halo-tile kernels (one product per tap: neither operand has a low part), several pairs
    per block, and the fp32-MFMA fallback for shapes outside the halo tiling (width 24, 14 x 14)."""
    P = Pair()
    g = torch.Generator().manual_seed(O * 3 + C)
    dz32, dz16 = P.acts(torch.randn(B, H, W, O, generator=g) * 2e-3)
    x32, x16 = P.acts(torch.randn(B, H, W, C, generator=g))
    amax = dz32.abs().max().reshape(1).contiguous()
    nws = P.lib.pp_conv3x3_bwd_weight_workspace(O, C, B, H, W)
    dws = []
    for K, d, xx in ((P.lib, dz32, x32), (P.lib16, dz16, x16)):
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
        dw = torch.zeros(O, C, 3, 3, device=dev())
        K.pp_conv3x3_bwd_weight_f16x3(d.data_ptr(), O, O, xx.data_ptr(), C, C, C, B, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nws,
                                      amax.data_ptr(), P.st)
        torch.cuda.synchronize()
        dws.append(dw)
    assert rel(dws[1], dws[0]) < TOL_F32
    ref = torch.nn.grad.conv2d_weight(x32.permute(0, 3, 1, 2).double(), (O, C, 3, 3), dz32.permute(0, 3, 1, 2).double(), padding=1)
    assert rel(dws[1], ref) < 1e-4          # and against the definition


# ------------------------------------------------------------------------------------------------ Winograd F(4x4,3x3)
@pytest.mark.parametrize('B,H,W,Cin,Cout,dil', [(2, 16, 16, 256, 256, 1), (2, 16, 16, 512, 64, 2), (4, 8, 12, 256, 128, 1)])
def test_winograd_paths(B, H, W, Cin, Cout, dil):
    """pp_conv3x3_wino_fwd_bn_h16, pp_conv3x3_wino_bwd_data_f16x3_h16 and
    pp_conv3x3_wino_bwd_weight_f16x3_h16 (kept transformed input and own transform)."""
    P = Pair()
    g = torch.Generator().manual_seed(Cin + Cout + dil)
    x32, x16 = P.acts(torch.randn(B, H, W, Cin, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dev())
    bias = torch.randn(Cout, generator=g).to(dev())
    Uf, Ub = torch.empty(36, Cout, Cin, device=dev()), torch.empty(36, Cin, Cout, device=dev())
    P.lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, 4, Uf.data_ptr(), Ub.data_ptr(), P.st)
    groups = 2
    nws = max(P.lib.pp_conv3x3_wino_workspace(Cin, Cout, B, H, W, dil), P.lib.pp_conv3x3_wino_workspace(Cout, Cin, B, H, W, dil),
              P.lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, H, W, dil)) + 256
    nvk = P.lib.pp_conv3x3_wino_vkeep_elems(Cin, B, H, W, dil)
    nst = groups * 2048 * 2 * Cout * 8
    res = []
    for K, xx, dt in ((P.lib, x32, torch.float32), (P.lib16, x16, KIND.dtype)):
        ws = torch.empty(nws, dtype=torch.uint8, device=dev())
        vk = torch.empty(nvk, device=dev())
        out = torch.zeros(B, H, W, Cout, device=dev(), dtype=dt)
        stats = torch.zeros(nst // 8 + 2, device=dev(), dtype=torch.float64)
        rows = ctypes.c_int(0)
        K.pp_conv3x3_wino_fwd_bn(xx.data_ptr(), Cin, Cin, Uf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, H, W, dil, 1,
                                 vk.data_ptr(), ws.data_ptr(), nws, 1, None, None, SLOPE, groups, stats.data_ptr(), nst, ctypes.byref(rows), P.st)
        torch.cuda.synchronize()
        tot = stats[:groups * rows.value * 2 * Cout].view(groups, rows.value, 2, Cout).sum(1).clone()
        res.append(dict(out=out, tot=tot, vk=vk, ws=ws))
    check_act(res[1]['out'], res[0]['out'], 'winograd z')
    assert rel(res[1]['tot'], res[0]['tot']) < TOL_F32
    assert torch.equal(res[1]['vk'][:-4], res[0]['vk'][:-4])        # the transformed input is computed in fp32 from the same values
    dz32, dz16 = P.acts(torch.randn(B, H, W, Cout, generator=g) * 2e-3)
    amax = dz32.abs().max().reshape(1).contiguous()
    dxs, dws, dws_own = [], [], []
    for (K, d, xx, dt), r in zip(((P.lib, dz32, x32, torch.float32), (P.lib16, dz16, x16, KIND.dtype)), res):
        dx = torch.zeros(B, H, W, Cin, device=dev(), dtype=dt)
        K.pp_conv3x3_wino_bwd_data_f16x3(d.data_ptr(), Cout, Cout, Ub.data_ptr(), dx.data_ptr(), Cin, Cin, B, H, W, dil, 0,
                                         r['ws'].data_ptr(), nws, amax.data_ptr(), P.st)
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev())
        K.pp_conv3x3_wino_bwd_weight_f16x3(d.data_ptr(), Cout, Cout, xx.data_ptr(), Cin, Cin, B, H, W, dil, dw.data_ptr(), 0,
                                           r['vk'].data_ptr(), r['ws'].data_ptr(), nws, amax.data_ptr(), P.st)
        dw2 = torch.zeros(Cout, Cin, 3, 3, device=dev())
        K.pp_conv3x3_wino_bwd_weight_f16x3(d.data_ptr(), Cout, Cout, xx.data_ptr(), Cin, Cin, B, H, W, dil, dw2.data_ptr(), 0,
                                           None, r['ws'].data_ptr(), nws, None, P.st)
        torch.cuda.synchronize()
        dxs.append(dx); dws.append(dw); dws_own.append(dw2)
    check_act(dxs[1], dxs[0], 'winograd dx')
    assert rel(dws[1], dws[0]) < TOL_F32 and rel(dws_own[1], dws_own[0]) < TOL_F32


# ------------------------------------------------------------------------------------------------ BatchNorm + LeakyReLU passes
@pytest.mark.parametrize('C,B,H,W,groups', [(32, 4, 16, 16, 2), (64, 2, 8, 12, 1), (12, 2, 6, 10, 2)])
def test_batchnorm_passes(C, B, H, W, groups):
    """Statistics, apply (+ fused pooling), the backward passes (+ fused pool gradient, eval-mode forms, split sums / apply) and
    pp_lazy_materialize: fp16 tensors in, fp32 arithmetic, fp16 tensors out."""
    from pacingpseudo_amd._lib import PpLazyIn
    P = Pair()
    g = torch.Generator().manual_seed(C + B)
    ppg = (B // groups) * H * W
    z32, z16 = P.acts(torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev()), (torch.randn(C, generator=g) * 0.2).to(dev())
    nws = P.lib.pp_bn_workspace(C, ppg, groups) + 12 * groups * C * 8 + 4096
    st32, st16 = {}, {}
    for K, z, S in ((P.lib, z32, st32), (P.lib16, z16, st16)):
        S['rm'], S['rv'] = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
        S['nbt'] = torch.zeros((), dtype=torch.int64, device=dev())
        S['coef'] = torch.zeros(4, groups, C, device=dev())
        S['ws'] = torch.empty(nws, dtype=torch.uint8, device=dev())
        mean, invstd, scale, shift = (S['coef'][i].data_ptr() for i in range(4))
        K.pp_bn_train_stats(z.data_ptr(), C, C, ppg, groups, 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), S['rm'].data_ptr(),
                            S['rv'].data_ptr(), S['nbt'].data_ptr(), mean, invstd, scale, shift, S['ws'].data_ptr(), nws, P.st)
        torch.cuda.synchronize()
    assert rel(st16['coef'], st32['coef']) < TOL_F32 and rel(st16['rv'], st32['rv']) < TOL_F32
    # from here on both forms use the SAME coefficients (the fp32 form's), so that only the passes themselves are compared
    coef = st32['coef']
    mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
    y32, y16 = torch.zeros(B, H, W, C, device=dev()), torch.zeros(B, H, W, C, device=dev(), dtype=KIND.dtype)
    P.call('pp_bn_lrelu_fwd', (z32.data_ptr(), C, scale, shift, y32.data_ptr(), C, C, ppg, groups, SLOPE),
           (z16.data_ptr(), C, scale, shift, y16.data_ptr(), C, C, ppg, groups, SLOPE))
    check_act(y16, y32, 'y')
    if H % 2 == 0 and W % 2 == 0:
        yp32, yp16 = torch.zeros_like(y32), torch.zeros_like(y16)
        pl32 = torch.zeros(B, H // 2, W // 2, C, device=dev())
        pl16 = torch.zeros(B, H // 2, W // 2, C, device=dev(), dtype=KIND.dtype)
        P.call('pp_bn_lrelu_fwd_pool', (z32.data_ptr(), C, scale, shift, yp32.data_ptr(), C, pl32.data_ptr(), C, C, B, H, W, groups, SLOPE),
               (z16.data_ptr(), C, scale, shift, yp16.data_ptr(), C, pl16.data_ptr(), C, C, B, H, W, groups, SLOPE))
        check_act(yp16, yp32, 'y (pool form)')
        check_act(pl16, pl32, 'pooled')
        assert torch.equal(pl16, torch.nn.functional.max_pool2d(yp16.permute(0, 3, 1, 2).float(), 2, 2).permute(0, 2, 3, 1).to(KIND.dtype))
    dy32, dy16 = P.acts(torch.randn(B, H, W, C, generator=g) * 1e-2)
    dp32, dp16 = P.acts(torch.randn(B, H // 2, W // 2, C, generator=g) * 1e-2)

    def grads():
        return [torch.zeros(C, device=dev()) for _ in range(3)]
    for training in (1, 0):
        out = []
        for K, dy, z, dt, S in ((P.lib, dy32, z32, torch.float32, st32), (P.lib16, dy16, z16, KIND.dtype, st16)):
            dz = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
            gg, gbeta, gb = grads()
            am = torch.zeros(1, device=dev())
            K.pp_bn_lrelu_bwd_amax(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(), training,
                                   dz.data_ptr(), C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg, groups, SLOPE,
                                   S['ws'].data_ptr(), nws, am.data_ptr(), P.st)
            dz2 = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
            gg2, gbeta2, gb2 = grads()
            K.pp_bn_lrelu_bwd(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(), training,
                              dz2.data_ptr(), C, gg2.data_ptr(), gbeta2.data_ptr(), gb2.data_ptr(), 0, C, ppg, groups, SLOPE,
                              S['ws'].data_ptr(), nws, P.st)
            torch.cuda.synchronize()
            out.append((dz, gg, gbeta, gb, am, dz2, gg2))
        a, b = out
        check_act(b[0], a[0], f'dz (training={training})')
        check_act(b[5], a[5], f'dz, no amax (training={training})')
        for i in (1, 2, 3, 6):
            assert rel(b[i], a[i]) < TOL_F32, (training, i)
        assert rel(b[4], a[4]) < 1e-3                       # max |dz|: taken before the rounding in both forms
        if H % 2 == 0 and W % 2 == 0:
            outp = []
            for K, dy, dpool, z, dt, S in ((P.lib, dy32, dp32, z32, torch.float32, st32), (P.lib16, dy16, dp16, z16, KIND.dtype, st16)):
                dz = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
                gg, gbeta, gb = grads()
                am = torch.zeros(1, device=dev())
                K.pp_bn_lrelu_bwd_pool(dy.data_ptr(), C, dpool.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(),
                                       training, dz.data_ptr(), C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, B, H, W, groups,
                                       SLOPE, S['ws'].data_ptr(), nws, am.data_ptr(), P.st)
                torch.cuda.synchronize()
                outp.append((dz, gg, gbeta, gb))
            check_act(outp[1][0], outp[0][0], f'dz with pool gradient (training={training})')
            for i in (1, 2, 3):
                assert rel(outp[1][i], outp[0][i]) < TOL_F32
    # eval-mode one-pass backward from y (+ pool gradient)
    oute = []
    for K, dy, dpool, y, dt, S in ((P.lib, dy32, dp32, y32, torch.float32, st32), (P.lib16, dy16, dp16, y16, KIND.dtype, st16)):
        yy = r16(y32).to(dtype=dt)                              # the same y values on both sides
        dz = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
        gg, gbeta, gb = grads()
        sc1 = coef[2, 0].contiguous()
        K.pp_bn_lrelu_bwd_eval(dy.data_ptr(), C, yy.data_ptr(), C, sc1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dz.data_ptr(), C,
                               gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, B * H * W, SLOPE, S['ws'].data_ptr(), nws, None, P.st)
        dzp = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
        ggp, gbetap, gbp = grads()
        if H % 2 == 0 and W % 2 == 0:
            K.pp_bn_lrelu_bwd_eval_pool(dy.data_ptr(), C, dpool.data_ptr(), C, yy.data_ptr(), C, sc1.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), dzp.data_ptr(), C, ggp.data_ptr(), gbetap.data_ptr(), gbp.data_ptr(), 0, C, B, H, W,
                                        SLOPE, S['ws'].data_ptr(), nws, None, P.st)
        torch.cuda.synchronize()
        oute.append((dz, gg, gbeta, dzp, ggp))
    check_act(oute[1][0], oute[0][0], 'dz (eval form)')
    check_act(oute[1][3], oute[0][3], 'dz (eval form with pool gradient)')
    assert rel(oute[1][1], oute[0][1]) < TOL_F32 and rel(oute[1][4], oute[0][4]) < TOL_F32
    # split statistics (synchronised BatchNorm): per-channel sums, backward sums + apply
    sums = []
    for K, z, dy, dt, S in ((P.lib, z32, dy32, torch.float32, st32), (P.lib16, z16, dy16, KIND.dtype, st16)):
        s1 = torch.zeros(groups, 2, C, device=dev(), dtype=torch.float64)
        K.pp_bn_stats_sums(z.data_ptr(), C, C, ppg, groups, s1.data_ptr(), S['ws'].data_ptr(), nws, P.st)
        s2 = torch.zeros(groups, 2, C, device=dev(), dtype=torch.float64)
        K.pp_bn_lrelu_bwd_sums(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, C, ppg, groups, SLOPE, s2.data_ptr(),
                               S['ws'].data_ptr(), nws, P.st)
        dz = torch.zeros(B, H, W, C, device=dev(), dtype=dt)
        gg, gbeta, gb = grads()
        K.pp_bn_lrelu_bwd_apply(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(), 1, s2.data_ptr(),
                                s2.data_ptr(), ppg, dz.data_ptr(), C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg, groups,
                                SLOPE, S['ws'].data_ptr(), nws, None, P.st)
        torch.cuda.synchronize()
        sums.append((s1, s2, dz, gg))
    assert rel(sums[1][0], sums[0][0]) < 1e-9 and rel(sums[1][1], sums[0][1]) < 1e-6
    check_act(sums[1][2], sums[0][2], 'dz (split form)')
    # lazy tensor -> values
    lz_rows = torch.stack([coef[2], coef[3], torch.full((groups, C), SLOPE, device=dev())], 1).contiguous()
    lz = PpLazyIn(lz_rows.data_ptr(), C, groups)
    m32, m16 = torch.zeros(B, H, W, C, device=dev()), torch.zeros(B, H, W, C, device=dev(), dtype=KIND.dtype)
    P.call('pp_lazy_materialize', (z32.data_ptr(), C, ctypes.byref(lz), m32.data_ptr(), C, C, B, H * W),
           (z16.data_ptr(), C, ctypes.byref(lz), m16.data_ptr(), C, C, B, H * W))
    check_act(m16, m32, 'materialised y')
    assert torch.equal(m16, y16)                              # the lazy form IS the apply pass


# ------------------------------------------------------------------------------------------------ pooling, resizing, 1x1 head, copies
@pytest.mark.parametrize('C,N,H,W,groups', [(32, 4, 16, 16, 2), (12, 2, 6, 10, 1)])
def test_spatial_entry_points(C, N, H, W, groups):
    from pacingpseudo_amd._lib import PpLazyIn
    P = Pair()
    g = torch.Generator().manual_seed(C * 3 + N)
    c0, ld = 8, C + 16                                 # a channel slice of a wider buffer, as the engine passes them
    es = {torch.float32: 4, KIND.dtype: 2}
    buf = torch.randn(N, H, W, ld, generator=g)
    b32, b16 = P.acts(buf)
    coef = torch.stack([torch.rand(groups, ld, generator=g) + 0.5, torch.randn(groups, ld, generator=g) * 0.3,
                        torch.full((groups, ld), SLOPE)], 1).to(dev()).contiguous()
    lz = PpLazyIn(coef.data_ptr() + 4 * c0, ld, groups)
    img = torch.randn(N, 1, H, W, generator=g).to(dev())
    K_cls = 5
    w = (torch.randn(K_cls, C, generator=g) / math.sqrt(C)).to(dev())
    bias = torch.randn(K_cls, generator=g).to(dev())
    dl = (torch.randn(N, K_cls, H, W, generator=g) * 1e-2).to(dev())
    dp32, dp16 = P.acts(torch.randn(N, H // 2, W // 2, C, generator=g) * 1e-2)
    du32, du16 = P.acts(torch.randn(N, 2 * H, 2 * W, C, generator=g) * 1e-2)
    msk = (torch.rand(N, C, generator=g) > 0.3).float().to(dev()) / 0.7
    nws = P.lib.pp_conv1x1_bwd_workspace(K_cls, C, N, H * W)
    R = []
    for K, b, dpool, dup, dt in ((P.lib, b32, dp32, du32, torch.float32), (P.lib16, b16, dp16, du16, KIND.dtype)):
        view = b.data_ptr() + es[dt] * c0
        r = {}

        def z(*shape):
            return torch.zeros(*shape, device=dev(), dtype=dt)
        r['packed'] = z(N, H, W, 4)
        K.pp_pack_image_nchw_to_nhwc(img.data_ptr(), N, 1, H, W, r['packed'].data_ptr(), 4, 4, P.st)
        r['pool'] = z(N, H // 2, W // 2, C)
        K.pp_maxpool2_fwd(view, ld, r['pool'].data_ptr(), C, C, N, H, W, P.st)
        r['dpx'] = z(N, H, W, C)
        K.pp_maxpool2_bwd(view, ld, dpool.data_ptr(), C, r['dpx'].data_ptr(), C, C, N, H, W, 0, P.st)
        r['up'] = z(N, 2 * H, 2 * W, C)
        K.pp_bilinear_fwd(view, ld, r['up'].data_ptr(), C, C, N, H, W, 2 * H, 2 * W, P.st)
        r['dup'] = (torch.ones(N, H, W, C, device=dev()) * 0.25).to(dt)
        K.pp_bilinear_bwd(dup.data_ptr(), C, r['dup'].data_ptr(), C, C, N, H, W, 2 * H, 2 * W, 1, P.st)
        r['copy'] = (torch.ones(N, H, W, C + 4, device=dev()) * 0.5).to(dt)
        K.pp_copy_slab(view, ld, r['copy'].data_ptr(), C + 4, C, N * H * W, 1, P.st)
        r['scaled'] = z(N, H, W, C)
        K.pp_channel_scale(view, ld, r['scaled'].data_ptr(), C, msk.data_ptr(), C, N, H * W, 0, P.st)
        r['logits'], r['logits_l'] = torch.zeros(N, K_cls, H, W, device=dev()), torch.zeros(N, K_cls, H, W, device=dev())
        K.pp_conv1x1_nhwc_to_nchw_fwd(view, ld, C, w.data_ptr(), bias.data_ptr(), r['logits'].data_ptr(), K_cls, N, H * W, P.st)
        K.pp_conv1x1_nhwc_to_nchw_fwd_lazy(view, ld, C, w.data_ptr(), bias.data_ptr(), r['logits_l'].data_ptr(), K_cls, N, H * W,
                                           ctypes.byref(lz), P.st)
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev())
        r['dx'], r['dx_l'] = z(N, H, W, C), z(N, H, W, C)
        r['dw'], r['db'] = torch.zeros(K_cls, C, device=dev()), torch.zeros(K_cls, device=dev())
        r['dw_l'], r['db_l'] = torch.zeros(K_cls, C, device=dev()), torch.zeros(K_cls, device=dev())
        K.pp_conv1x1_nchw_to_nhwc_bwd(dl.data_ptr(), view, ld, C, w.data_ptr(), r['dx'].data_ptr(), C, r['dw'].data_ptr(),
                                      r['db'].data_ptr(), K_cls, N, H * W, 0, 0, ws.data_ptr(), nws, P.st)
        K.pp_conv1x1_nchw_to_nhwc_bwd_lazy(dl.data_ptr(), view, ld, C, w.data_ptr(), r['dx_l'].data_ptr(), C, r['dw_l'].data_ptr(),
                                           r['db_l'].data_ptr(), K_cls, N, H * W, 0, 0, ws.data_ptr(), nws, ctypes.byref(lz), P.st)
        torch.cuda.synchronize()
        R.append(r)
    a, b = R
    for k in ('packed', 'pool', 'dpx', 'up', 'dup', 'copy', 'scaled', 'dx', 'dx_l'):
        check_act(b[k], a[k], k)
    assert torch.equal(b['pool'].float(), a['pool'])            # a maximum of fp16 numbers is one of them: exact
    assert torch.equal(b['dpx'].float(), a['dpx'])              # routing of fp16 gradients: exact
    for k in ('logits', 'logits_l', 'dw', 'db', 'dw_l', 'db_l'):
        assert rel(b[k], a[k]) < TOL_F32, k
    assert torch.equal(b16[..., :c0], r16(buf[..., :c0]).to(KIND.dtype).to(dev()))      # the neighbours of the view are untouched


def test_memory_update_reads_fp16_features():
    P = Pair()
    g = torch.Generator().manual_seed(3)
    hid, h, w, K, H, W = 64, 8, 8, 5, 32, 32
    f32, f16 = P.acts(torch.randn(1, h, w, hid, generator=g))
    scb = torch.zeros(K + 1, H, W)
    scb[torch.randint(0, K + 1, (H, W), generator=g), torch.arange(H)[:, None], torch.arange(W)[None, :]] = 1.0
    scb = scb.to(dev())
    for cosine in (0, 1):
        b32, b16 = P.bufs(torch.randn(K, hid, generator=g) * (1.0 if cosine else 0.0))
        ws = torch.empty(P.lib.pp_memory_update_workspace(K, hid), dtype=torch.uint8, device=dev())
        P.lib.pp_memory_update(f32.data_ptr(), hid, hid, h, w, scb.data_ptr(), K, H, W, b32.data_ptr(), 0.9, cosine, ws.data_ptr(), ws.numel(), P.st)
        getattr(P.lib, 'pp_memory_update' + KIND.suffix)(f16.data_ptr(), hid, hid, h, w, scb.data_ptr(), K, H, W, b16.data_ptr(), 0.9, cosine,
                                                         ws.data_ptr(), ws.numel(), P.st)
        torch.cuda.synchronize()
        assert torch.equal(b16, b32)                            # same values in, same arithmetic


# ------------------------------------------------------------------------------------------------ the whole step
# Stated tolerances of the 16-bit storage mode as a whole (one training step of the benchmark network at random initial weights,
# both BatchNorm modes), against the fp32 path of the same library on the same weights and batch:
# (fp16 / bf16: 11 / 8 significand bits -- a stored tensor is rounded by 2^-12 / 2^-9 relative, and the whole-step bounds scale with it)
TOL_LOGITS = {'fp16': 3e-2, 'bf16': 2e-1}       # max-norm relative error of the logits
TOL_LOSS = {'fp16': 5e-3, 'bf16': 4e-2}         # absolute; the losses are O(1)
MIN_COSINE = {'fp16': 0.9, 'bf16': 0.6}         # every parameter gradient against its fp32 counterpart (fp16 measured 0.96-1.0, see the report)
TOL_HEAD_GRAD = {'fp16': 2e-2, 'bf16': 1.5e-1}  # relative L2 of the head's weight gradient (no LeakyReLU decision between it and the loss)
MIN_ARGMAX = {'fp16': 0.99, 'bf16': 0.90}       # share of pixels whose arg-max class agrees with the fp32 storage mode (random initial weights: near-tie logits; bf16 measured 0.937)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize('size,num_classes,bn_eval', [(256, 5, False), (224, 2, False), (256, 5, True)])
def test_training_step_in_16_bit_storage(size, num_classes, bn_eval):
    """One full-flags step (weak + strong pass, auxiliary path, memory bank) at full channel widths with `--storage fp16` against
    the fp32 storage mode: forward inside the stated tolerance, every gradient finite and aligned with its fp32 counterpart, and
    the fraction of LeakyReLU branch decisions that differ between the two forwards -- a rounded pre-activation within 2^-11 of
    the kink takes the other branch, which changes that element's gradient by a factor 100: THE source of the gradient
    difference (relative L2 ~ sqrt(fraction) per layer, accumulating towards the encoder), not the 16-bit gradient tensors.
    224 x 224 / 2 classes is the LVSC geometry of BASELINE config 5 (widths that are not multiples of 32 below full
    resolution: the general split-fp16 kernels and the fp32-MFMA weight-gradient fallback run)."""
    from oracle import pacing_oracle as O
    from tests import _golden as G
    from tests.test_gpu_step import build_model
    ign = num_classes
    a32 = O.full_flags(num_classes=num_classes, ignored_index=ign)
    a16 = O.full_flags(num_classes=num_classes, ignored_index=ign)
    a16.storage = KIND.name
    torch.manual_seed(1)
    m32 = build_model(a32)
    m16 = build_model(a16, {k: v.detach().cpu().numpy() for k, v in m32.state_dict().items()})
    assert m16.engine.h16 and not m32.engine.h16
    batch = {k: v.cuda() for k, v in O.synthetic_batch(2, size, size, num_classes=num_classes, seed=11, keep=0.03).items() if k != 'label'}
    rec = {}
    for name, m in (('fp32', m32), ('h16', m16)):
        m.train()
        if bn_eval:
            m.eval()                      # the reference's state from epoch 1 on: BatchNorm with running statistics
        out = m(batch, mode='train', step=0)
        loss = sum(out[k] for k in ('loss_pce', 'loss_ent', 'loss_cr', 'loss_aux_cls', 'loss_memory'))
        loss.backward()
        torch.cuda.synchronize()
        eng = m.engine
        masks = {L.name: eng.branch_mask(L).clone() for L in eng.layers}
        rec[name] = dict(out={k: v.detach().float().clone() for k, v in out.items()},
                         grads={n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None}, masks=masks)
        assert eng.last_plan.h16 == (name == 'h16') and eng.last_plan.act_dtype == (KIND.dtype if name == 'h16' else torch.float32)
    a, b = rec['fp32'], rec['h16']
    errs = {}
    for k in ('segmentation/logits', 'segmentation/logits_strong', 'logits_aux_cls'):
        errs[k] = rel(b['out'][k], a['out'][k])
        assert errs[k] < TOL_LOGITS[KIND.name], (k, errs[k])
        assert errs[k] > 1e-5, f'{k}: {errs[k]:.1e} is fp32 grade -- the 16-bit kernels did not run'
    for k in ('loss_pce', 'loss_ent', 'loss_cr', 'loss_aux_cls', 'loss_memory'):
        assert abs(float(b['out'][k]) - float(a['out'][k])) < TOL_LOSS[KIND.name], k
    agree = float((b['out']['segmentation/logits'].argmax(1) == a['out']['segmentation/logits'].argmax(1)).float().mean())
    assert agree > MIN_ARGMAX[KIND.name], agree
    flips = {n: float((a['masks'][n] != b['masks'][n]).float().mean()) for n in a['masks']}
    cosines, rels = {}, {}
    for n, ga in a['grads'].items():
        gb = b['grads'][n]
        assert bool(torch.isfinite(gb).all()), n
        if float(ga.norm()) == 0.0:
            continue
        cosines[n] = float(torch.dot(ga.flatten().double(), gb.flatten().double()) / (ga.double().norm() * gb.double().norm() + 1e-300))
        rels[n] = float((ga - gb).double().norm() / ga.double().norm())
        assert cosines[n] > MIN_COSINE[KIND.name], (n, cosines[n])
    # the head sees no branch decision between itself and the loss: its gradient is 16-bit-rounding accurate
    assert rels['backbone.final_conv.weight'] < TOL_HEAD_GRAD[KIND.name], rels['backbone.final_conv.weight']
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
    G._report(dict(kind='storage_' + KIND.name, tag=f'{num_classes}-class {size}x{size} full width, bn_eval={bn_eval}',
                   tolerance_logits=TOL_LOGITS[KIND.name], logits_rel_err=errs, argmax_agreement=agree,
                   leaky_relu_branch_flip_fraction=dict(max=max(flips.values()), mean=sum(flips.values()) / len(flips)),
                   gradient_cosine=dict(min=min(cosines.values()), head=cosines['backbone.final_conv.weight']),
                   gradient_rel_l2=dict(worst=worst, head=rels['backbone.final_conv.weight']),
                   loss_scale=m16.engine.loss_scale))


def test_bare_unet_in_16_bit_storage():
    """The backbone by itself (`UNet.forward` -> `_UNetFunction`: the gradient of the logits arrives from torch's autograd, is
    multiplied by the loss scale on its way into the plan and the slab is divided by it afterwards) against the fp32 storage mode."""
    from types import SimpleNamespace
    from pacingpseudo_amd.engine import StepEngine
    from pacingpseudo_amd.models import UNet
    kw = dict(input_ch=1, init_ch=32, max_ch=512, num_classes=4, output_stride=16, is_stride_conv=False, is_trans_conv=False,
              elab_end_points=False)
    torch.manual_seed(3)
    n32 = UNet(**kw).cuda()
    n16 = UNet(**kw).cuda()
    n16.load_state_dict(n32.state_dict())
    n16._engine = StepEngine(n16, None, SimpleNamespace(storage=KIND.name))
    x = torch.randn(2, 1, 128, 128, generator=torch.Generator().manual_seed(1)).cuda()
    tgt = torch.randn(2, 4, 128, 128, generator=torch.Generator().manual_seed(2)).cuda()
    res = []
    for net in (n32, n16):
        net.train()
        out = net(x)['segmentation/logits']
        loss = ((out - tgt) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}))
    assert n16._engine.last_plan.h16 and not n32._engine.last_plan.h16
    (o32, g32), (o16, g16) = res
    assert 1e-5 < rel(o16, o32) < TOL_LOGITS[KIND.name]
    for n, g in g32.items():
        assert bool(torch.isfinite(g16[n]).all()), n
        if float(g.norm()) > 0:
            cos = float(torch.dot(g.flatten().double(), g16[n].flatten().double()) / (g.double().norm() * g16[n].double().norm() + 1e-300))
            assert cos > MIN_COSINE[KIND.name], (n, cos)
    gh32, gh16 = g32['final_conv.weight'], g16['final_conv.weight']
    assert float((gh32 - gh16).norm() / gh32.norm()) < TOL_HEAD_GRAD[KIND.name]          # also proves the loss scale is gone from the slab


def test_loss_scale_overflow_skips_the_update():
    """A static loss scale can overflow fp16 in a bad step.  The unscaling pass (pp_scale_guard) then raises the slab's flag, the
    fused optimizer leaves weights and moments untouched and counts the skipped update (GradScaler.step semantics); the next
    good step trains again.  Forced here with an absurd scale."""
    from oracle import pacing_oracle as O
    from pacingpseudo_amd.optim import FusedAdam
    from tests.test_gpu_step import build_model
    if KIND.name == 'bf16':
        pytest.skip('bfloat16 has the exponent range of fp32: a loss scale cannot overflow the stored gradients (the guard is the same code)')
    a16 = O.full_flags()
    a16.storage = KIND.name
    torch.manual_seed(1)
    m = build_model(a16)
    m.engine.loss_scale = 2.0 ** 40                       # read when the plan is built
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=0.0)
    batch = {k: v.cuda() for k, v in O.synthetic_batch(2, 128, 128, seed=3, keep=0.05).items() if k != 'label'}
    m.train()

    def step():
        out = m(batch, mode='train', step=0)
        loss = sum(out[k] for k in ('loss_pce', 'loss_ent', 'loss_cr', 'loss_aux_cls', 'loss_memory'))
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
    before = m.flat.params.clone()
    step()
    assert int(m.flat.guard[0]) == 1 and int(m.flat.guard[1]) == 1          # flagged; ONE skipped step (two segments, one count)
    assert torch.equal(m.flat.params, before)
    st = next(iter(opt._slabs.values()))
    assert float(st['m'].abs().max()) == 0.0 and float(st['v'].abs().max()) == 0.0
    assert opt.state_dict()['slabs'][0]['steps'] == {}     # Adam's step counts (device side) did not advance: bias corrections stay in step
    # a sane scale for the next plan: the same model trains
    m.engine.loss_scale = 1024.0
    m.engine.plans.clear()
    skipped = int(m.flat.guard[1])
    step()
    assert int(m.flat.guard[0]) == 0 and int(m.flat.guard[1]) == skipped
    assert not torch.equal(m.flat.params, before) and bool(torch.isfinite(m.flat.params).all())
    assert opt.state_dict()['slabs'][0]['steps'] == {'backbone': 1, 'aux_path': 1}      # the first update that really happened
