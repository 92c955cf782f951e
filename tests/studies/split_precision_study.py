"""Numerical study (CPU, no GPU): can the convolution GEMMs run on 16-bit MFMAs with split operands and still meet the
1e-4 fp32-parity bar?  Every 3x3 convolution of the oracle's UNet forward is replaced by an emulation of

  * fp32          : plain fp32 conv (what v_mfma_f32_32x32x2_f32 computes)
  * wino_fp32     : Winograd F(4x4,3x3), fp32 transforms, fp32 GEMM  (the shipped wide-layer path)
  * bf16x3        : x = hi + lo in bf16,  hi*hi + hi*lo + lo*hi, fp32 accumulation
  * fp16x3        : x = hi + lo/2048 in fp16 (lo pre-scaled by 2^11), same three products, fp32 accumulation
  * wino_fp16x3   : the fp16x3 split applied to the Winograd-domain operands V and U

and the logits are compared with an fp64 run of the same network.  Products of two 11-bit (fp16) or 8-bit (bf16)
significands are exact in fp32, so an fp32 conv over the rounded tensors IS what the 16-bit MFMA would accumulate.
Usage: python tests/studies/split_precision_study.py [size=64] [batch=2]"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pacing_oracle as O

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(0)
args = O.default_args()
sd = O.init_state(args, seed=1)
x = torch.randn(B, 1, S, S)

G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]])
At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])


def split16(t, dtype, scale):
    hi = t.to(dtype).to(t.dtype)
    lo = ((t - hi) * scale).to(dtype).to(t.dtype)
    return hi, lo


def mm_split(a, b, dtype, scale):            # a [.., M, K] @ b [.., K, N] with split operands, fp32 accumulate
    ah, al = split16(a, dtype, scale)
    bh, bl = split16(b, dtype, scale)
    return ah @ bh + (ah @ bl + al @ bh) / scale


def conv_split(x, w, b, dil, dtype, scale):
    xh, xl = split16(x, dtype, scale)
    wh, wl = split16(w, dtype, scale)
    y = F.conv2d(xh, wh, None, 1, dil, dil) + (F.conv2d(xh, wl, None, 1, dil, dil) + F.conv2d(xl, wh, None, 1, dil, dil)) / scale
    return y + b.view(1, -1, 1, 1)


def conv_wino(x, w, b, dil, mm):
    if dil > 1:                                # a dilation-d conv = d*d independent dilation-1 convs on x[i::d, j::d]
        y = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3], dtype=x.dtype)
        for i in range(dil):
            for j in range(dil):
                y[:, :, i::dil, j::dil] = conv_wino(x[:, :, i::dil, j::dil].contiguous(), w, b, 1, mm)
        return y
    N, C, H, W = x.shape
    if H % 4 or W % 4:
        return F.conv2d(x, w, b, 1, 1, 1)
    dt = x.dtype
    g, bt, at = G.to(dt), Bt.to(dt), At.to(dt)
    U = torch.einsum('ar,ocrs,bs->abco', g, w, g)                       # [6,6,C,O]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 6, 4).unfold(3, 6, 4)                          # [N,C,th,tw,6,6]
    V = torch.einsum('ar,ncyxrs,bs->abnyxc', bt, tiles, bt)             # [6,6,N,th,tw,C]
    th, tw = V.shape[3], V.shape[4]
    M = mm(V.reshape(6, 6, N * th * tw, C), U)                          # [6,6,T,O]
    Y = torch.einsum('ar,rstO,bs->tOab', at, M, at)                     # [T,O,4,4]
    Y = Y.reshape(N, th, tw, -1, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(N, -1, H, W)
    return Y + b.view(1, -1, 1, 1)


MODES = {
    'fp32': lambda x, w, b, d: F.conv2d(x, w, b, 1, d, d),
    'wino_fp32': lambda x, w, b, d: conv_wino(x, w, b, d, lambda a, u: a @ u) if w.shape[1] >= 128 else F.conv2d(x, w, b, 1, d, d),
    'bf16x3': lambda x, w, b, d: conv_split(x, w, b, d, torch.bfloat16, 256.0),
    'fp16x3': lambda x, w, b, d: conv_split(x, w, b, d, torch.float16, 2048.0),
    'wino_fp16x3': lambda x, w, b, d: (conv_wino(x, w, b, d, lambda a, u: mm_split(a, u, torch.float16, 2048.0))
                                       if w.shape[1] >= 128 else conv_split(x, w, b, d, torch.float16, 2048.0)),
}

orig = F.conv2d


def run(mode, dtype):
    sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    fn = MODES.get(mode)

    def patched(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if fn is None or weight.shape[-1] != 3 or patched.busy:
            return orig(inp, weight, bias, stride, padding, dilation, groups)
        patched.busy = True
        try:
            d = dilation if isinstance(dilation, int) else dilation[0]
            return fn(inp, weight, bias, d)
        finally:
            patched.busy = False
    patched.busy = False
    O.F.conv2d = patched
    try:
        with torch.no_grad():
            return O.unet_forward(sdd, x.to(dtype), args, True)['segmentation/logits']
    finally:
        O.F.conv2d = orig


ref = run(None, torch.float64)
print(f'UNet forward, batch {B}, {S}x{S}, train-mode BN; max |logits - fp64| (logits range {ref.abs().max():.2f})')
for mode in MODES:
    out = run(mode, torch.float32)
    err = (out.double() - ref).abs()
    print(f'  {mode:12s} max {err.max():.3e}   mean {err.mean():.3e}')
