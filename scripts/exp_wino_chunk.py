"""Experiment (GPU box): does running a Winograd layer in batch chunks keep V / M in the 256 MB Infinity Cache?
Times pp_conv3x3_wino_fwd_f16x3 / bwd_data_f16x3 / bwd_weight_f16x3 for 64 images in one call vs in chunks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr
LAYERS = {'enc4c2': (256, 256, 32, 1), 'enc5c2': (512, 512, 32, 2), 'dec5c1': (1024, 512, 32, 1), 'dec4c1': (768, 256, 32, 1), 'dec3c1': (384, 128, 64, 1)}
B = 64
dev = torch.device('cuda', 0); st = stream_ptr()
def timeit(f, n=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, (Cin, Cout, S, dil) in LAYERS.items():
    x = torch.randn(B, S, S, Cin, device=dev); dz = torch.randn(B, S, S, Cout, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    Uf = torch.empty(36, Cout, Cin, device=dev); Ub = torch.empty(36, Cin, Cout, device=dev)
    lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, 4, Uf.data_ptr(), Ub.data_ptr(), st)
    out = torch.empty(B, S, S, Cout, device=dev); dx = torch.empty(B, S, S, Cin, device=dev); dw = torch.empty_like(w)
    nws = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, S, S, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, S, S, dil),
              lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, S, S, dil))
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
    vk = torch.empty(lib.pp_conv3x3_wino_vkeep_elems(Cin, B, S, S, dil), device=dev)
    for chunk in (64, 32, 16, 8):
        def fwd(keep):
            for b0 in range(0, B, chunk):
                xs, os_ = x[b0:b0 + chunk], out[b0:b0 + chunk]
                vkp = None
                if keep:
                    per = lib.pp_conv3x3_wino_vkeep_elems(Cin, chunk, S, S, dil)
                    vkp = vk.data_ptr() + 4 * per * (b0 // chunk) if per * (B // chunk) <= vk.numel() else None
                lib.pp_conv3x3_wino_fwd_f16x3(xs.data_ptr(), Cin, Cin, Uf.data_ptr(), bias.data_ptr(), os_.data_ptr(), Cout, Cout, chunk, S, S, dil, 0, vkp, ws.data_ptr(), nws, st)
        def dgrad():
            for b0 in range(0, B, chunk):
                lib.pp_conv3x3_wino_bwd_data_f16x3(dz[b0:b0 + chunk].data_ptr(), Cout, Cout, Ub.data_ptr(), dx[b0:b0 + chunk].data_ptr(), Cin, Cin, chunk, S, S, dil, 0, ws.data_ptr(), nws, None, st)
        t_f, t_fk, t_d = timeit(lambda: fwd(False)), timeit(lambda: fwd(True)), timeit(dgrad)
        print(f'{name:7s} chunk {chunk:3d}: fwd {t_f:7.3f} ms  fwd+keep {t_fk:7.3f} ms  dgrad {t_d:7.3f} ms', flush=True)
