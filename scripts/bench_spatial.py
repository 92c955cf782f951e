"""Per-launch time and algorithmic bandwidth of the HBM-bound spatial family at the benchmark's shapes (GPU box):
    python scripts/bench_spatial.py [images]      (images = weak + strong views, default 64)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0); st = stream_ptr()


def timed(name, f, nbytes, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f'{name:34s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic', flush=True)


# 1x1 head (dec1 output 32 channels at 256^2 -> 5 classes) and its backward
C, K, S = 32, 5, 256
x = torch.randn(B, S, S, C, device=dev); w = torch.randn(K, C, device=dev); b = torch.randn(K, device=dev)
logits = torch.empty(B, K, S, S, device=dev); dl = torch.randn(B, K, S, S, device=dev)
dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
nws = lib.pp_conv1x1_bwd_workspace(K, C, B, S * S); ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
P = B * S * S
timed('conv1x1 head fwd 32->5 @256', lambda: lib.pp_conv1x1_nhwc_to_nchw_fwd(x.data_ptr(), C, C, w.data_ptr(), b.data_ptr(), logits.data_ptr(), K, B, S * S, st), 4.0 * P * (C + K))
timed('conv1x1 head bwd 32->5 @256', lambda: lib.pp_conv1x1_nchw_to_nhwc_bwd(dl.data_ptr(), x.data_ptr(), C, C, w.data_ptr(), dx.data_ptr(), C, dw.data_ptr(), db.data_ptr(), K, B, S * S, 0, 0, ws.data_ptr(), nws, st), 4.0 * P * (2 * C + K))
del x, dx, logits, dl
# bilinear x2 into / out of the [up | skip] concat buffers of decoder stages 1..3
for name, Cl, Cs, Si in (('dec1', 64, 32, 128), ('dec2', 128, 64, 64), ('dec3', 256, 128, 32)):
    lo = torch.randn(B, Si, Si, Cl, device=dev); cat = torch.empty(B, 2 * Si, 2 * Si, Cl + Cs, device=dev); glo = torch.empty_like(lo)
    nb = 4.0 * B * Cl * (Si * Si + 4 * Si * Si)
    timed(f'bilinear fwd {name} {Cl}ch {Si}->{2 * Si}', lambda: lib.pp_bilinear_fwd(lo.data_ptr(), Cl, cat.data_ptr(), Cl + Cs, Cl, B, Si, Si, 2 * Si, 2 * Si, st), nb)
    timed(f'bilinear bwd {name} {Cl}ch {2 * Si}->{Si}', lambda: lib.pp_bilinear_bwd(cat.data_ptr(), Cl + Cs, glo.data_ptr(), Cl, Cl, B, Si, Si, 2 * Si, 2 * Si, 0, st), nb)
    del lo, cat, glo
# 2x2 max-pool of encoder stages 1..3 (output = skip slice of the concat buffer) and its backward (accumulating)
for name, Cc, Cw, Si in (('enc1', 32, 96, 256), ('enc2', 64, 192, 128), ('enc3', 128, 384, 64)):
    cat = torch.randn(B, Si, Si, Cw, device=dev); pooled = torch.empty(B, Si // 2, Si // 2, Cc, device=dev)
    dp = torch.randn_like(pooled); gcat = torch.zeros_like(cat)
    off = (Cw - Cc) * 4
    timed(f'maxpool fwd {name} {Cc}ch @{Si}', lambda: lib.pp_maxpool2_fwd(cat.data_ptr() + off, Cw, pooled.data_ptr(), Cc, Cc, B, Si, Si, st), 4.0 * B * Cc * Si * Si * 1.25)
    timed(f'maxpool bwd {name} {Cc}ch @{Si}', lambda: lib.pp_maxpool2_bwd(cat.data_ptr() + off, Cw, dp.data_ptr(), Cc, gcat.data_ptr() + off, Cw, Cc, B, Si, Si, 1, st), 4.0 * B * Cc * Si * Si * 3.25)
    del cat, pooled, dp, gcat
