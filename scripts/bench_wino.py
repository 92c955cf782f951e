"""Winograd vs direct convolution per layer (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr
LAYERS = {'enc3c2': (128, 128, 64, 1), 'enc4c1': (128, 256, 32, 1), 'enc4c2': (256, 256, 32, 1),
          'enc5c1': (256, 512, 32, 2), 'enc5c2': (512, 512, 32, 2), 'enc6c1': (512, 512, 32, 4),
          'dec5c1': (1024, 512, 32, 1), 'dec4c1': (768, 256, 32, 1), 'dec3c1': (384, 128, 64, 1), 'aux': (1024, 64, 32, 1),
          'dec2c1': (192, 64, 128, 1), 'enc2c2': (64, 64, 128, 1)}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0); st = stream_ptr()
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f'{"layer":8s} {"op":6s} {"direct ms":>10s} {"wino ms":>9s} {"speedup":>8s} {"alg TF/s":>9s}')
for name, (Cin, Cout, S, dil) in LAYERS.items():
    x = torch.randn(B, S, S, Cin, device=dev); dz = torch.randn(B, S, S, Cout, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    wf = torch.empty(Cout, 9, Cin, device=dev); wb = torch.empty(Cin, 9, Cout, device=dev)
    lib.pp_pack_conv3x3_weights(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), wb.data_ptr(), st)
    tile = lib.pp_conv3x3_wino_tile(S, S, dil); planes = (tile + 2) ** 2
    Uf = torch.empty(planes, Cout, Cin, device=dev); Ub = torch.empty(planes, Cin, Cout, device=dev)
    lib.pp_wino_pack_weights(w.data_ptr(), Cout, Cin, tile, Uf.data_ptr(), Ub.data_ptr(), st)
    out = torch.empty(B, S, S, Cout, device=dev); dx = torch.empty(B, S, S, Cin, device=dev); dw = torch.empty_like(w)
    n1 = lib.pp_conv3x3_bwd_weight_workspace(Cout, Cin, B, S, S)
    n2 = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, S, S, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, S, S, dil),
             lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, S, S, dil))
    ws = torch.empty(max(n1, n2) + 64, dtype=torch.uint8, device=dev); nws = max(n1, n2)
    flops = 2.0 * B * S * S * 9 * Cin * Cout
    pairs = {
        'fwd': (lambda: lib.pp_conv3x3_fwd(x.data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, st),
                lambda: lib.pp_conv3x3_wino_fwd(x.data_ptr(), Cin, Cin, Uf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, None, ws.data_ptr(), nws, st)),
        'dgrad': (lambda: lib.pp_conv3x3_bwd_data(dz.data_ptr(), Cout, Cout, wb.data_ptr(), dx.data_ptr(), Cin, Cin, B, S, S, dil, 0, st),
                  lambda: lib.pp_conv3x3_wino_bwd_data(dz.data_ptr(), Cout, Cout, Ub.data_ptr(), dx.data_ptr(), Cin, Cin, B, S, S, dil, 0, ws.data_ptr(), nws, st)),
        'wgrad': (lambda: lib.pp_conv3x3_bwd_weight(dz.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, Cin, B, S, S, dil, dw.data_ptr(), 0, ws.data_ptr(), nws, st),
                  lambda: lib.pp_conv3x3_wino_bwd_weight(dz.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, B, S, S, dil, dw.data_ptr(), 0, None, ws.data_ptr(), nws, st)),
    }
    f16 = {}
    if tile == 4:
        Uf16 = torch.empty_like(Uf); Ub16 = torch.empty_like(Ub)
        lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, tile, Uf16.data_ptr(), Ub16.data_ptr(), st)
        f16 = {'fwd': lambda: lib.pp_conv3x3_wino_fwd_f16x3(x.data_ptr(), Cin, Cin, Uf16.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, None, ws.data_ptr(), nws, st),
               'wgrad': lambda: lib.pp_conv3x3_wino_bwd_weight_f16x3(dz.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, B, S, S, dil, dw.data_ptr(), 0, None, ws.data_ptr(), nws, None, st),
               'dgrad': lambda: lib.pp_conv3x3_wino_bwd_data_f16x3(dz.data_ptr(), Cout, Cout, Ub16.data_ptr(), dx.data_ptr(), Cin, Cin, B, S, S, dil, 0, ws.data_ptr(), nws, None, st)}
    for op, (fd, fw) in pairs.items():
        td, tw = timeit(fd), timeit(fw)
        extra = ''
        if op in f16:
            t16 = timeit(f16[op])
            extra = f'   wino f16x3 {t16:7.3f} ms  {flops / t16 / 1e9:7.1f} alg TF/s'
        print(f'{name:8s} {op:6s} {td:10.3f} {tw:9.3f} {td / tw:8.2f} {flops / tw / 1e9:9.1f}{extra}')
