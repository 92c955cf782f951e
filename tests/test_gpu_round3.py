"""Round-3 GPU cases: the mixed-precision mode of BASELINE config 5 (`--precision fp16`) at the LVSC geometry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests.test_gpu_step import build_model, iteration  # noqa: E402

# Stated tolerances of the fp16-operand mode (11 significand bits per operand, fp32 accumulation, fp32 tensors in HBM):
TOL_MIXED_LOGITS = 3e-2     # max-norm relative error of the logits against the fp32 oracle
TOL_MIXED_LOSS = 2e-2       # absolute, losses are O(1)


class _Fp16Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


@pytest.mark.timeout(1200)
def test_mixed_precision_step_at_the_lvsc_geometry():
    """BASELINE.json configs[4]: LVSC (2 classes, 224 x 224 crops), full flags, full channel widths, mixed precision.  One
    training step with the forward / data-gradient products of the halo-tile and Winograd kernels on fp16 operands
    (pp_set_matrix_products(1)) against (a) the fp32 oracle, with the stated mixed-precision tolerance, and (b) the oracle
    with both operands of every 3x3 convolution rounded to fp16 -- the same input rounding the direct kernels apply -- to
    show that the difference IS operand rounding and nothing else.  The fp32 mode of the same process afterwards is back
    inside 1e-4."""
    from pacingpseudo_amd._lib import lib
    from pacingpseudo_amd.optim import FusedAdam
    args = O.full_flags(num_classes=2, ignored_index=2)
    torch.manual_seed(1)
    model = build_model(args)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = O.synthetic_batch(2, 224, 224, num_classes=2, seed=11, keep=0.03)
    torch.set_num_threads(min(32, max(torch.get_num_threads(), 8)))
    ref_out, ref_grads, ref_total = O.train_step({k: v.clone() for k, v in sd.items()}, batch, 0, args, training=True)
    O.CONV_OPERAND_ROUND = lambda t, prefix: _Fp16Round.apply(t)
    try:
        rnd_out, _, rnd_total = O.train_step({k: v.clone() for k, v in sd.items()}, batch, 0, args, training=True)
    finally:
        O.CONV_OPERAND_ROUND = None
    assert lib.pp_get_matrix_products() == 3
    lib.pp_set_matrix_products(1)
    try:
        opt = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
        rec, grads = iteration(model, opt, batch, args, 0)
    finally:
        lib.pp_set_matrix_products(3)
    report = {}
    for key in ('segmentation/logits', 'segmentation/logits_strong', 'logits_aux_cls'):
        got = rec[key].double().cpu().numpy()
        e32, e16 = G.rel_err(got, ref_out[key].numpy()), G.rel_err(got, rnd_out[key].numpy())
        base = G.rel_err(rnd_out[key].numpy(), ref_out[key].numpy())          # what the rounding alone does to the oracle
        report[key] = (e32, e16, base)
        assert e32 < TOL_MIXED_LOGITS, (key, e32)
        assert e32 > 1e-4, f'{key}: {e32:.1e} is fp32 grade -- the fp16-operand kernels did not run'
        assert e32 < 4 * base + 1e-3, (key, e32, base)                          # the size operand rounding explains
    for key in ('loss_pce', 'loss_ent', 'loss_cr', 'loss_aux_cls', 'loss_memory'):
        assert abs(float(rec[key]) - float(ref_out[key])) < TOL_MIXED_LOSS * max(1.0, abs(float(ref_out[key]))), key
    agree = (rec['segmentation/logits'].argmax(1).cpu() == ref_out['segmentation/logits'].argmax(1)).float().mean()
    assert float(agree) > 0.99, float(agree)
    # gradients: finite, and pointing where the fp32 gradients point
    cosines = {}
    for k in ('backbone.dec_block1.conv_block.conv_layer2.conv.weight', 'backbone.enc_block5.conv_block.conv_layer2.conv.weight',
              'backbone.dec_block5.conv_block.conv_layer1.conv.weight', 'backbone.final_conv.weight'):
        a, b = grads[k].double().cpu().flatten(), ref_grads[k].double().flatten()
        assert bool(torch.isfinite(a).all())
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        cosines[k] = cos
        assert cos > 0.9, (k, cos)          # 2-image scribble-sparse batch: measured 0.97-0.999 (first run: enc5.c2 0.968)
    G._report(dict(kind='mixed_precision', tag='2-class 224x224 full width, fp16 operands', tolerance_logits=TOL_MIXED_LOGITS,
                   errors={k: dict(vs_fp32_oracle=v[0], vs_fp16_rounded_oracle=v[1], rounding_alone_in_the_oracle=v[2]) for k, v in report.items()},
                   argmax_agreement=float(agree), gradient_cosine_vs_fp32_oracle=cosines))
    # back in the default mode the same model / batch is inside the fp32 tolerance again (mode is per process, switchable)
    model2 = build_model(args, {k: v.numpy() for k, v in sd.items()})
    with torch.no_grad():
        out32 = model2({k: v.cuda() for k, v in batch.items() if k != 'label'}, mode='train', step=0)
    assert G.rel_err(out32['segmentation/logits'].double().cpu().numpy(), ref_out['segmentation/logits'].numpy()) < 1e-4
