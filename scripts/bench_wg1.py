"""Micro-benchmark (GPU box): BatchNorm backward of the first layer with the weight gradient folded in against the separate
launches, benchmark shape (32 channels, 2 x 32 images of 256 x 256).   python scripts/bench_wg1.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pacingpseudo_amd._lib import lib, stream_ptr
dev = torch.device('cuda', 0); st = stream_ptr()
C, B, H, W, G = 32, 32, 256, 256, 2
N = B * G
x = torch.randn(N, H, W, 4, device=dev); z = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev) * 1e-3
dz = torch.empty(N, H, W, C, device=dev)
coef = torch.rand(4, G, C, device=dev) + 0.5
mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
ppg = B * H * W
nws = max(lib.pp_bn_lrelu_bwd_wgrad_c1_workspace(C, ppg, G), lib.pp_bn_lrelu_bwd_wgrad_c1_workspace(C, N * H * W, 1), lib.pp_conv3x3_bwd_weight_workspace(C, 4, N, H, W))
ws = torch.empty(nws + 64, dtype=torch.uint8, device=dev)
dw = torch.empty(C, 1, 3, 3, device=dev); dg, db, dbc = (torch.empty(C, device=dev) for _ in range(3))
def fused(): lib.pp_bn_lrelu_bwd_wgrad_c1(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(), 1, x.data_ptr(), 4, H, W, dw.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, ppg, G, 0.01, ws.data_ptr(), nws, st)
def fused_eval(): lib.pp_bn_lrelu_bwd_eval_wgrad_c1(dy.data_ptr(), C, z.data_ptr(), C, scale, gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), 4, H, W, dw.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, N * H * W, 0.01, ws.data_ptr(), nws, st)
def separate():
    lib.pp_bn_lrelu_bwd(dy.data_ptr(), C, z.data_ptr(), C, scale, shift, mean, invstd, gamma.data_ptr(), 1, dz.data_ptr(), C, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, ppg, G, 0.01, ws.data_ptr(), nws, st)
    lib.pp_conv3x3_bwd_weight(dz.data_ptr(), C, C, x.data_ptr(), 4, 4, 1, N, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nws, st)
def separate_eval():
    lib.pp_bn_lrelu_bwd_eval(dy.data_ptr(), C, z.data_ptr(), C, scale, gamma.data_ptr(), beta.data_ptr(), dz.data_ptr(), C, dg.data_ptr(), db.data_ptr(), dbc.data_ptr(), 0, C, N * H * W, 0.01, ws.data_ptr(), nws, None, st)
    lib.pp_conv3x3_bwd_weight(dz.data_ptr(), C, C, x.data_ptr(), 4, 4, 1, N, H, W, 1, dw.data_ptr(), 0, ws.data_ptr(), nws, st)
for name, f in (('separate', separate), ('fused', fused), ('separate_eval', separate_eval), ('fused_eval', fused_eval)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:14s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us', flush=True)
