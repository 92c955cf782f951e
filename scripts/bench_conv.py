"""Micro-benchmark of the convolution kernels per layer shape (GPU box).  Usage:
   python scripts/bench_conv.py [--batch 64] [--iters 5] [--only fwd,dgrad,wgrad] [--layers enc1c2,dec5c1,...]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr

LAYERS = {  # name: (Cin, Cout, HW, dil)
    'enc1c1': (1, 32, 256, 1), 'enc1c2': (32, 32, 256, 1), 'enc2c1': (32, 64, 128, 1), 'enc2c2': (64, 64, 128, 1),
    'enc3c1': (64, 128, 64, 1), 'enc3c2': (128, 128, 64, 1), 'enc4c1': (128, 256, 32, 1), 'enc4c2': (256, 256, 32, 1),
    'enc5c1': (256, 512, 32, 2), 'enc5c2': (512, 512, 32, 2), 'enc6c1': (512, 512, 32, 4),
    'dec5c1': (1024, 512, 32, 1), 'dec4c1': (768, 256, 32, 1), 'dec3c1': (384, 128, 64, 1), 'dec2c1': (192, 64, 128, 1),
    'dec1c1': (96, 32, 256, 1), 'aux': (1024, 64, 32, 1),
}
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--only', default='fwd,dgrad,wgrad')
ap.add_argument('--layers', default=','.join(LAYERS))
ap.add_argument('--f16x3', action='store_true', help='time the split-fp16 fwd / dgrad entry points')
a = ap.parse_args()
dev = torch.device('cuda', 0)
st = stream_ptr()
print(f'{"layer":8s} {"op":6s} {"GFLOP":>8s} {"ms":>8s} {"TFLOP/s":>8s}   variant={os.environ.get("PP_CONV_VARIANT", "auto")}/{os.environ.get("PP_WGRAD_VARIANT", "auto")}')
for name in a.layers.split(','):
    Cin, Cout, S, dil = LAYERS[name]
    B = a.batch
    ipad = (Cin + 3) // 4 * 4
    x = torch.randn(B, S, S, ipad, device=dev)
    dz = torch.randn(B, S, S, Cout, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    wf = torch.empty(Cout, 9, ipad, device=dev); wb = torch.empty(Cin, 9, Cout, device=dev)
    lib.pp_pack_conv3x3_weights(w.data_ptr(), Cout, Cin, ipad, wf.data_ptr(), wb.data_ptr() if ipad == Cin else None, st)
    if a.f16x3:
        lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, ipad, wf.data_ptr(), wb.data_ptr() if ipad == Cin else None, st)
    out = torch.empty(B, S, S, Cout, device=dev)
    dx = torch.empty(B, S, S, ipad, device=dev)
    dw = torch.empty_like(w)
    nws = lib.pp_conv3x3_bwd_weight_workspace(Cout, ipad, B, S, S)
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
    flops = 2.0 * B * S * S * 9 * ipad * Cout
    amax = dz.abs().max().reshape(1)
    ops = {
        'fwd': lambda: lib.pp_conv3x3_fwd(x.data_ptr(), ipad, ipad, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, st),
        'dgrad': (lambda: lib.pp_conv3x3_bwd_data(dz.data_ptr(), Cout, Cout, wb.data_ptr(), dx.data_ptr(), ipad, Cin, B, S, S, dil, 0, st)) if ipad == Cin else None,
        'fwd16': lambda: lib.pp_conv3x3_fwd_f16x3(x.data_ptr(), ipad, ipad, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, None, st),
        'dgrad16': (lambda: lib.pp_conv3x3_bwd_data_f16x3(dz.data_ptr(), Cout, Cout, wb.data_ptr(), dx.data_ptr(), ipad, Cin, B, S, S, dil, 0, amax.data_ptr(), st)) if ipad == Cin else None,
        'wgrad16': lambda: lib.pp_conv3x3_bwd_weight_f16x3(dz.data_ptr(), Cout, Cout, x.data_ptr(), ipad, ipad, Cin, B, S, S, dil, dw.data_ptr(), 0, ws.data_ptr(), nws, amax.data_ptr(), st),
        'wgrad': lambda: lib.pp_conv3x3_bwd_weight(dz.data_ptr(), Cout, Cout, x.data_ptr(), ipad, ipad, Cin, B, S, S, dil, dw.data_ptr(), 0, ws.data_ptr(), nws, st),
    }
    for op in a.only.split(','):
        if a.f16x3 and op in ('fwd', 'dgrad', 'wgrad'):
            op += '16'
        f = ops.get(op)
        if f is None:
            continue
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(f'{name:8s} {op:6s} {flops / 1e9:8.1f} {ms:8.3f} {flops / ms / 1e9:8.1f}')
