"""Winograd weight-gradient GEMM time against the number of reduction splits, per layer of the benchmark net (GPU box):
    python scripts/sweep_wino_wg_splits.py [images]
The library's own events time the GEMM family alone (transforms and the finalize are separate families)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr, prof_collect
LAYERS = {'enc4c2': (256, 256, 32, 1), 'enc5c1': (256, 512, 32, 2), 'enc5c2': (512, 512, 32, 2), 'dec5c1': (1024, 512, 32, 1),
          'dec4c1': (768, 256, 32, 1), 'dec3c1': (384, 128, 64, 1)}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0); st = stream_ptr()
for name, (Cin, Cout, S, dil) in LAYERS.items():
    x = torch.randn(B, S, S, Cin, device=dev); dz = torch.randn(B, S, S, Cout, device=dev) * 1e-4
    am = dz.abs().max().reshape(1).contiguous()
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    vk = torch.empty(lib.pp_conv3x3_wino_vkeep_elems(Cin, B, S, S, dil), device=dev)
    row = []
    for sp in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32):
        if sp:
            os.environ['PP_WINO_WG_SPLITS'] = str(sp)
        else:
            os.environ.pop('PP_WINO_WG_SPLITS', None)
        nws = lib.pp_conv3x3_wino_bwd_weight_workspace(Cout, Cin, B, S, S, dil)
        ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
        f = lambda: lib.pp_conv3x3_wino_bwd_weight_f16x3(dz.data_ptr(), Cout, Cout, x.data_ptr(), Cin, Cin, B, S, S, dil, dw.data_ptr(), 0, None, ws.data_ptr(), nws, am.data_ptr(), st)
        for _ in range(2): f()
        torch.cuda.synchronize()
        lib.pp_prof_enable(1); prof_collect()
        n = 8
        for _ in range(n): f()
        torch.cuda.synchronize()
        lib.pp_prof_enable(0)
        pr = prof_collect()
        g = pr['wino_wgrad_f16x3']['ms'] / n
        tot = sum(v['ms'] for v in pr.values()) / n
        row.append(f'{sp if sp else "auto"}:{g * 1e3:.0f}/{tot * 1e3:.0f}')
        del ws
    print(f'{name:7s} GEMM us / whole call us   ' + '  '.join(row), flush=True)
