"""STUDY (round 6): a direct 3x3 convolution fed by LDS-DMA from an activation tensor stored as [hi | lo] fp16
pairs (pacingpseudo_amd/csrc/study/pp_study_pair.hip, libpp_study.so) against the shipped split-fp16 kernels of the same layers
(pp_conv3x3_fwd_f16x3: the two-half halo kernel), forward pass, the benchmark's batch of 64 images per launch.

    python tests/studies/pair_layout_study.py [--reps 20]        (GPU box)        -> one JSON line per layer

Both kernels execute the same three fp16 MFMA products per fp32 product with the same packed weights; the study kernel stages
nothing through registers and converts nothing (its input is pre-split), keeps all eight waves multiplying and synchronises once
per 54 MFMA steps.  The output is compared with the product kernel's and, on a small batch, with torch in fp64."""
import argparse
import ctypes as C
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pacingpseudo_amd._lib import lib, stream_ptr  # noqa: E402

LAYERS = [  # name, Cin, Cout, H = W
    ('enc1.c2 / dec1.c2  32->32 @256', 32, 32, 256),
    ('enc2.c1            32->64 @128', 32, 64, 128),
    ('enc2.c2 / dec2.c2  64->64 @128', 64, 64, 128),
    ('enc3.c1            64->128 @64', 64, 128, 64),
    ('dgrad of enc2.c1   64->32 @128', 64, 32, 128),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64)
    a = ap.parse_args()
    so = C.CDLL(os.path.join(ROOT, 'pacingpseudo_amd', 'lib', 'libpp_study.so'))
    so.pp_study_last_error.restype = C.c_char_p
    so.pp_study_split_pairs.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    so.pp_study_conv3x3_pair_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device('cuda', 0)
    st = stream_ptr()

    def timed(fn, reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3          # us

    for name, Cin, Cout, S in LAYERS:
        g = torch.Generator().manual_seed(Cin * 1000 + Cout)
        B = a.batch
        x = torch.randn(B, S, S, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).to(dev)
        bias = torch.randn(Cout, generator=g).to(dev)
        wf, wb = torch.empty(Cout, 9, Cin, device=dev), torch.empty(Cin, 9, Cout, device=dev)
        lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), wb.data_ptr(), st)
        xp = torch.empty_like(x)                       # the pair layout has the bytes of the fp32 tensor
        y_prod, y_pair = torch.empty(B, S, S, Cout, device=dev), torch.zeros(B, S, S, Cout, device=dev)

        def split():
            rc = so.pp_study_split_pairs(x.data_ptr(), Cin, B * S * S, xp.data_ptr(), st)
            assert rc == 0, so.pp_study_last_error()

        def prod():
            lib.pp_conv3x3_fwd_f16x3(x.data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), y_prod.data_ptr(), Cout, Cout, B, S, S, 1, 0, None, st)

        def pair(mode=0, dst=None):
            rc = so.pp_study_conv3x3_pair_fwd(xp.data_ptr(), Cin, wf.data_ptr(), bias.data_ptr(), (dst if dst is not None else y_pair).data_ptr(), Cout, Cout,
                                              B, S, S, 0, mode, st)
            assert rc == 0, so.pp_study_last_error()

        split()
        prod()
        pair()
        torch.cuda.synchronize()
        scale = float(y_prod.abs().max())
        err_prod = float((y_pair - y_prod).abs().max()) / scale
        # fp64 reference on two images
        ref = F.conv2d(x[:2].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
        err_ref = float((y_pair[:2].double().cpu() - ref).abs().max() / ref.abs().max())
        err_ref_prod = float((y_prod[:2].double().cpu() - ref).abs().max() / ref.abs().max())
        t_split, t_prod, t_pair = timed(split, a.reps), timed(prod, a.reps), timed(pair, a.reps)
        # timing-only variants of the study kernel (wrong results by construction, written to a scratch tensor): what it is bound by
        scratch = torch.empty_like(y_pair)
        variants = {name_: round(timed(lambda m=m: pair(m, scratch), a.reps), 1)
                    for name_, m in (('no_mfma', 1), ('no_fragment_reads', 2), ('no_mfma_no_reads', 3), ('no_dma', 4), ('barriers_and_stores_only', 7))}
        flops = 2.0 * B * S * S * Cin * Cout * 9 * 3           # executed: three fp16 products per fp32 product
        byts = 4.0 * B * S * S * (Cin + Cout)
        print(json.dumps(dict(layer=name, batch=B, us_product_kernel=round(t_prod, 1), us_pair_dma_kernel=round(t_pair, 1),
                              speedup=round(t_prod / t_pair, 3), us_split_pass=round(t_split, 1), us_timing_only_variants=variants,
                              pflops_executed=dict(product=round(flops / t_prod / 1e9, 3), pair=round(flops / t_pair / 1e9, 3)),
                              tb_per_s_algorithmic=dict(product=round(byts / t_prod / 1e6, 2), pair=round(byts / t_pair / 1e6, 2)),
                              max_rel_err=dict(pair_vs_product=err_prod, pair_vs_fp64=err_ref, product_vs_fp64=err_ref_prod))), flush=True)
        assert err_ref < 1e-4, (name, err_ref)


if __name__ == '__main__':
    main()
