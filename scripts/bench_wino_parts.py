"""Per-family time of the split-fp16 Winograd calls of one layer (GPU box): transforms vs GEMM, from the library's events.
    python scripts/bench_wino_parts.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr, prof_collect
LAYERS = {'enc4c2': (256, 256, 32, 1), 'enc5c2': (512, 512, 32, 2), 'dec5c1': (1024, 512, 32, 1), 'dec4c1': (768, 256, 32, 1),
          'dec3c1': (384, 128, 64, 1), 'aux': (1024, 64, 32, 1)}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0); st = stream_ptr()
for name, (Cin, Cout, S, dil) in LAYERS.items():
    x = torch.randn(B, S, S, Cin, device=dev); dz = torch.randn(B, S, S, Cout, device=dev) * 1e-4
    am = dz.abs().max().reshape(1).contiguous()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    Uf = torch.empty(36, Cout, Cin, device=dev); Ub = torch.empty(36, Cin, Cout, device=dev)
    lib.pp_wino_pack_weights_f16x3(w.data_ptr(), Cout, Cin, 4, Uf.data_ptr(), Ub.data_ptr(), st)
    out = torch.empty(B, S, S, Cout, device=dev); dx = torch.empty(B, S, S, Cin, device=dev); dw = torch.empty_like(w)
    nws = max(lib.pp_conv3x3_wino_workspace(Cin, Cout, B, S, S, dil), lib.pp_conv3x3_wino_workspace(Cout, Cin, B, S, S, dil),
              lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, S, S, dil))
    ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
    vk = torch.empty(lib.pp_conv3x3_wino_vkeep_elems(Cin, B, S, S, dil), device=dev)
    ops = {
        'fwd': lambda: lib.pp_conv3x3_wino_fwd_f16x3(x.data_ptr(), Cin, Cin, Uf.data_ptr(), bias.data_ptr(), out.data_ptr(), Cout, Cout, B, S, S, dil, 0, vk.data_ptr(), ws.data_ptr(), nws, st),
        'dgrad': lambda: lib.pp_conv3x3_wino_bwd_data_f16x3(dz.data_ptr(), Cout, Cout, Ub.data_ptr(), dx.data_ptr(), Cin, Cin, B, S, S, dil, 0, ws.data_ptr(), nws, am.data_ptr(), st),
        'wgrad': lambda: lib.pp_conv3x3_wino_bwd_weight_f16x3(dz.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, B, S, S, dil, dw.data_ptr(), 0, vk.data_ptr(), ws.data_ptr(), nws, am.data_ptr(), st),
    }
    for op, f in ops.items():
        for _ in range(3): f()
        torch.cuda.synchronize()
        lib.pp_prof_enable(1); prof_collect()
        n = 10
        for _ in range(n): f()
        torch.cuda.synchronize()
        lib.pp_prof_enable(0)
        pr = prof_collect()
        parts = {k: v for k, v in pr.items() if v['launches']}
        msg = '  '.join(f"{k} {v['ms'] / n:6.3f} ms ({v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['flops'] else v['bytes'] / (v['ms'] * 1e-3) / 1e12:6.1f} {'TF/s' if v['flops'] else 'TB/s'})" for k, v in parts.items())
        print(f'{name:7s} {op:5s}  {msg}', flush=True)
