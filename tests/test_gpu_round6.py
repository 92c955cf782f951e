"""Round-6 GPU cases: the eval-mode BatchNorm coefficient rows of the whole backbone in one launch, and the gradient buckets of the
data-parallel step issued from the second stream.

Reference semantics: models/unet.py:189 (BatchNorm2d in eval mode: y = gamma (x - running_mean) / sqrt(running_var + eps) + beta),
the state the reference trains in from epoch 1 on (train_chaos.py:370)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests.test_gpu_step import build_model  # noqa: E402


def _steps(model, opt, batch, n, eval_from=1):
    outs = []
    model.train()
    for i in range(n):
        if i == eval_from:
            model.eval()
        out = model(batch, mode='train', step=1 if i >= eval_from else 0)
        loss = out['loss_pce'] + out['loss_ent'] * 0.1 + out['loss_cr'] * 0.1 + out['loss_aux_cls'] + out['loss_memory']
        opt.zero_grad()
        loss.backward()
        opt.step()
        outs.append({k: v.detach().clone() for k, v in out.items()})
    torch.cuda.synchronize()
    return outs


def test_eval_mode_coefficient_rows_in_one_launch_are_bit_identical():
    """pp_bn_eval_coeffs_batch (one launch per forward for all backbone layers) against the per-layer pp_bn_eval_coeffs calls it
    replaces: outputs of every step, parameters and BatchNorm buffers after one train-mode and three eval-mode steps equal bit for
    bit; the per-entry rows equal as well (two groups)."""
    from pacingpseudo_amd import engine as E
    from pacingpseudo_amd._lib import PpBnCoefItem, lib, stream_ptr
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags(init_ch=16, max_ch=64, hid_ch=16, feat_ch=[64, 64])
    batch = {k: v.cuda() for k, v in O.synthetic_batch(2, 64, 64, seed=5, keep=0.05).items() if k != 'label'}
    res = {}
    saved = E.COEF_BATCH
    try:
        for flag in (True, False):
            E.COEF_BATCH = flag
            torch.manual_seed(1)
            model = build_model(args)
            opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=args.wd)
            outs = _steps(model, opt, batch, 4)
            res[flag] = (outs, model.flat.params.clone(), {k: v.clone() for k, v in model.state_dict().items()})
            if flag:
                assert len(model.engine._coefs_ready) == len(model.engine.layers)      # the batched path really ran (eval forward)
    finally:
        E.COEF_BATCH = saved
    for a, b in zip(res[True][0], res[False][0]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(res[True][1], res[False][1])
    for k, v in res[True][2].items():
        assert torch.equal(v, res[False][2][k]), k
    # per entry: three layers of different width, two statistics groups
    g = torch.Generator().manual_seed(3)
    st = stream_ptr()
    items, outs_b, outs_s = [], [], []
    for C in (8, 64, 200):
        gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
        rm, rv = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.1).cuda()
        ob, os_ = torch.zeros(4, 2, C, device='cuda'), torch.zeros(4, 2, C, device='cuda')
        items.append((PpBnCoefItem(C, 2, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), *(ob[i].data_ptr() for i in range(4))),
                      (gamma, beta, rm, rv)))
        lib.pp_bn_eval_coeffs(C, 2, 1e-5, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), *(os_[i].data_ptr() for i in range(4)), st)
        outs_b.append(ob)
        outs_s.append(os_)
    arr = (PpBnCoefItem * len(items))(*(it for it, _ in items))
    lib.pp_bn_eval_coeffs_batch(arr, len(items), 1e-5, st)
    torch.cuda.synchronize()
    for ob, os_ in zip(outs_b, outs_s):
        assert torch.equal(ob, os_)


def test_pair_layout_study_kernel_equals_the_product_kernel():
    """The study kernel of DESIGN.md section 9 (libpp_study.so: a direct 3x3 convolution fed by LDS-DMA from an activation tensor stored
    as [hi | lo] fp16 pairs) executes the same split-fp16 products in the same order as the shipped halo kernel: bit-identical output,
    1e-6 of torch's fp64 convolution (models/unet.py:188), for both chunk counts, image borders and several tiles per block."""
    import ctypes as C
    import os
    import torch.nn.functional as F
    from pacingpseudo_amd._lib import lib, stream_ptr
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'pacingpseudo_amd', 'lib', 'libpp_study.so')
    if not os.path.exists(path):
        pytest.skip('libpp_study.so not built (make study)')
    so = C.CDLL(path)
    so.pp_study_last_error.restype = C.c_char_p
    so.pp_study_split_pairs.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    so.pp_study_conv3x3_pair_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    st = stream_ptr()
    for Cin, Cout, B, H, W, blocks_x in ((32, 32, 2, 16, 32, 0), (64, 64, 3, 24, 64, 5), (64, 96, 2, 8, 32, 1)):
        g = torch.Generator().manual_seed(Cin + Cout + H)
        x = torch.randn(B, H, W, Cin, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).cuda()
        bias = torch.randn(Cout, generator=g).cuda()
        wf, wb = torch.empty(Cout, 9, Cin, device='cuda'), torch.empty(Cin, 9, Cout, device='cuda')
        lib.pp_pack_conv3x3_weights_f16x3(w.data_ptr(), Cout, Cin, Cin, wf.data_ptr(), wb.data_ptr(), st)
        xp = torch.empty_like(x)
        assert so.pp_study_split_pairs(x.data_ptr(), Cin, B * H * W, xp.data_ptr(), st) == 0
        y_prod, y_pair = torch.empty(B, H, W, Cout, device='cuda'), torch.full((B, H, W, Cout), 7.0, device='cuda')
        lib.pp_conv3x3_fwd_f16x3(x.data_ptr(), Cin, Cin, wf.data_ptr(), bias.data_ptr(), y_prod.data_ptr(), Cout, Cout, B, H, W, 1, 0, None, st)
        rc = so.pp_study_conv3x3_pair_fwd(xp.data_ptr(), Cin, wf.data_ptr(), bias.data_ptr(), y_pair.data_ptr(), Cout, Cout, B, H, W, blocks_x, 0, st)
        assert rc == 0, so.pp_study_last_error()
        torch.cuda.synchronize()
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
        assert float((y_pair.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6
        assert torch.equal(y_pair, y_prod), (Cin, Cout, float((y_pair - y_prod).abs().max()))
