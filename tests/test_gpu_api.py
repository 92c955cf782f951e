"""Functional loss API (pacingpseudo_amd.losses.losses) and the training driver, on the GPU box."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacing_oracle as O  # noqa: E402
from tests import _golden as G  # noqa: E402


def test_functional_losses_match_reference_formulas():
    from pacingpseudo_amd.losses import losses as L
    g = torch.Generator().manual_seed(0)
    N, K, H, W = 2, 5, 16, 12
    zw = torch.randn(N, K, H, W, generator=g); zs = torch.randn(N, K, H, W, generator=g)
    t = torch.randint(0, K + 1, (N, H, W), generator=g)
    m = (torch.rand(N, 1, H, W, generator=g) > 0.4).float()
    cases = [
        (lambda a, b: L.partial_cross_entropy_loss(a, t.to(a.device), K), lambda a, b: O.partial_cross_entropy_loss(a, t, K), 'w'),
        (lambda a, b: L.entropy_minimization_loss(a, m.to(a.device)), lambda a, b: O.entropy_minimization_loss(a, m), 'w'),
        (lambda a, b: L.entropy_minimization_loss(a), lambda a, b: O.entropy_minimization_loss(a), 'w'),
        (lambda a, b: L.kl_loss(b, a, m.to(a.device)), lambda a, b: O.kl_loss(b, a, m), 'ws'),
        (lambda a, b: L.soft_label_cross_entropy_loss(b, torch.softmax(a, 1).detach(), m.to(a.device)),
         lambda a, b: O.soft_label_cross_entropy_loss(b, torch.softmax(a, 1).detach(), m), 's'),
        (lambda a, b: L.l2_loss(torch.softmax(b, 1), torch.softmax(a, 1).detach(), m.to(a.device)),
         lambda a, b: O.l2_loss(torch.softmax(b, 1), torch.softmax(a, 1).detach(), m), 's'),
        (lambda a, b: L.l1_loss(torch.softmax(b, 1), torch.softmax(a, 1).detach()),
         lambda a, b: O.l1_loss(torch.softmax(b, 1), torch.softmax(a, 1).detach()), 's'),
        # the target carries a gradient, as in the reference's call soft_label_cross_entropy_loss(strong, softmax(weak))
        # (consistency_reglur_memory.py:53-54; losses/losses.py:45-96 differentiate through `target`)
        (lambda a, b: L.soft_label_cross_entropy_loss(b, torch.softmax(a, 1), m.to(a.device)),
         lambda a, b: O.soft_label_cross_entropy_loss(b, torch.softmax(a, 1), m), 'ws'),
        (lambda a, b: L.soft_label_cross_entropy_loss(b, torch.softmax(a, 1)),
         lambda a, b: O.soft_label_cross_entropy_loss(b, torch.softmax(a, 1)), 'ws'),
        (lambda a, b: L.l2_loss(torch.softmax(b, 1), torch.softmax(a, 1), m.to(a.device)),
         lambda a, b: O.l2_loss(torch.softmax(b, 1), torch.softmax(a, 1), m), 'ws'),
        (lambda a, b: L.l1_loss(torch.softmax(b, 1), torch.softmax(a, 1)),
         lambda a, b: O.l1_loss(torch.softmax(b, 1), torch.softmax(a, 1)), 'ws'),
    ]
    for i, (mine, ref, wrt) in enumerate(cases):
        a, b = zw.cuda().requires_grad_(True), zs.cuda().requires_grad_(True)
        ar, br = zw.double().requires_grad_(True), zs.double().requires_grad_(True)
        lv = mine(a, b); rv = ref(ar, br)
        assert abs(float(lv) - float(rv)) < 1e-5 * max(1.0, abs(float(rv))), i
        lv.backward(); rv.backward()
        if 'w' in wrt:
            assert G.rel_err(a.grad.cpu().numpy(), ar.grad.numpy()) < 1e-4, i
        if 's' in wrt:
            assert G.rel_err(b.grad.cpu().numpy(), br.grad.numpy()) < 1e-4, i
    lg = torch.randn(5, 5, generator=g)
    tg = torch.arange(5)
    assert abs(float(L.cross_entropy_loss(lg.cuda(), tg.cuda())) - float(O.cross_entropy_loss(lg, tg))) < 1e-5


def test_bare_unet_inference_matches_oracle():
    from pacingpseudo_amd.models import UNet
    args = O.default_args(init_ch=8, max_ch=64)
    torch.manual_seed(3)
    net = UNet(input_ch=1, init_ch=8, max_ch=64, num_classes=5, output_stride=8, elab_end_points=True)
    sd = {'backbone.' + k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    x = torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ep = net(x.cuda())
        ref = O.unet_forward(sd, x, args, training=False)
    assert sorted(ep) == sorted(ref)
    for k in ref:
        assert G.rel_err(ep[k].cpu().numpy(), ref[k].numpy()) < 1e-4, k
    # with gradients enabled the same call is the fully-supervised trainer's forward (upper_bound_chaos.py:156):
    # the logits carry an autograd node (tests/test_gpu_round2.py::test_bare_unet_trains_like_the_upper_bound_reference)
    assert net(x.cuda())['segmentation/logits'].requires_grad


def test_training_driver_runs_and_writes_the_reference_artefacts(tmp_path):
    from pacingpseudo_amd.train import train_main
    root = str(tmp_path / 'out')
    vd = train_main(['--tag', 'smoke', '--session', 'Experiment', '--root', root, '--synthetic', '8', '--epoch', '2',
                     '--batch_size', '4', '--image_size', '64', '--num_workers', '0', '--cpu_input', '--do_loss_ent', '--do_decoder_consistency', '--do_aux_path', '--do_memory'])
    assert vd.shape == (2,) and np.isfinite(vd).all()
    run = glob.glob(os.path.join(root, 't1', 'Experiment', 'Experiment-*-fold1-smoke'))
    assert len(run) == 1
    for f in ('log.txt', 'valdice.npz', 'ckps/ckp_1.pth'):
        assert os.path.exists(os.path.join(run[0], f)), f
    sd = torch.load(os.path.join(run[0], 'ckps', 'ckp_1.pth'), map_location='cpu')
    assert len(sd) == 165 and 'aux_path.memory_bank' in sd and 'backbone.final_conv.weight' in sd
    assert int(sd['backbone.enc_block1.conv_block.conv_layer1.norm_op.num_batches_tracked']) == 2 * 2   # epoch 0 only
    log = open(os.path.join(run[0], 'log.txt')).read()
    assert 'epoch: 001, lr: ' in log and 'loss_memory' in log and 'val: 001' in log and 'All: ' in log
    # the scalar tags of the reference's TensorBoard writer (train_chaos.py:362-367, :416-423), one JSON line each
    import json
    rows = [json.loads(x) for x in open(os.path.join(run[0], 'tb_summary', 'scalars.jsonl'))]
    tags = {r['tag'] for r in rows}
    assert {'losses/loss_pce_train', 'losses/loss_cr', 'losses/loss_ent', 'losses/loss_aux_cls', 'losses/loss_memory', 'lr/current_lr',
            'losses/loss_pce_val', 'DSC/BG', 'DSC/Liver', 'DSC/R-Kidney', 'DSC/L-Kidney', 'DSC/Spleen', 'DSC/All', 'DSC/Best'} <= tags
    assert sorted({r['step'] for r in rows}) == [0, 1]
    assert abs([r['value'] for r in rows if r['tag'] == 'DSC/All'][-1] - vd[1]) < 1e-12


@pytest.mark.parametrize('recipe', ['TransformsColor', 'TransformsColorMixup'])
def test_training_driver_with_the_gpu_input_pipeline(tmp_path, recipe):
    """The default input path: raw slices from the loader, the reference's two-stream augmentation on the device (augment.py),
    the training step on its output; validation at the native phantom size.  ACDC preset (4 classes, 224 crop overridden to
    64 for speed; no modality level in the run directory, as for the reference's ACDC / LVSC outputs)."""
    from pacingpseudo_amd.train import train_main
    root = str(tmp_path / 'out')
    vd = train_main(['--tag', 'aug', '--session', 'Experiment', '--root', root, '--dataset', 'acdc', '--synthetic', '8',
                     '--epoch', '2', '--batch_size', '4', '--image_size', '64', '--num_workers', '0',
                     '--augmentations', recipe, '--do_loss_ent', '--do_decoder_consistency', '--do_aux_path', '--do_memory'])
    assert vd.shape == (2,) and np.isfinite(vd).all()
    run = glob.glob(os.path.join(root, 'Experiment', 'Experiment-*-fold1-aug'))
    assert len(run) == 1
    log = open(os.path.join(run[0], 'log.txt')).read()
    assert 'num_classes=4' in log and 'RV' in log and 'loss_cr' in log
    for line in log.splitlines():
        if 'loss_pce' in line and 'epoch:' in line:
            assert 'nan' not in line.lower(), line


def test_drivers_with_the_strided_transposed_variant(tmp_path):
    """--is_stride_conv / --is_trans_conv through the CLI exactly as the reference declares them (`type=bool`: any non-empty
    string is True, train_chaos.py:77-84): the training driver trains and validates, the checkpoint carries the transposed-convolution
    weights, and inference.py loads its backbone strictly into the same variant and scores the phantom test set."""
    from pacingpseudo_amd import inference as I
    from pacingpseudo_amd.train import train_main
    root = str(tmp_path / 'out')
    vd = train_main(['--tag', 'sc', '--session', 'Experiment', '--root', root, '--synthetic', '8', '--epoch', '2', '--batch_size', '4',
                     '--image_size', '64', '--num_workers', '0', '--output_stride', '16', '--is_stride_conv', 'True', '--is_trans_conv', 'True',
                     '--do_loss_ent', '--do_decoder_consistency', '--do_aux_path', '--do_memory'])
    assert vd.shape == (2,) and np.isfinite(vd).all()
    run = glob.glob(os.path.join(root, 't1', 'Experiment', 'Experiment-*-fold1-sc'))
    assert len(run) == 1
    sd = torch.load(os.path.join(run[0], 'ckps', 'ckp_1.pth'), map_location='cpu')
    assert len(sd) == 170 and sd['backbone.dec_block4.up_samp.weight'].shape == (512, 256, 2, 2)
    assert sd['backbone.dec_block5.up_samp.weight'].shape == (512, 512, 1, 1)
    log = open(os.path.join(run[0], 'log.txt')).read()
    assert 'is_stride_conv=True' in log and 'nan' not in log.lower().split('all:')[0][-2000:]
    dicearr, hd95arr = I.main(['--fold', '1', '--checkpoint_file', os.path.join(run[0], 'ckps', 'ckp_1.pth'), '--root', str(tmp_path / 'inf'),
                               '--dataset', 'chaost1', '--synthetic', '6', '--image_size', '64', '--batch_size', '4', '--num_workers', '0',
                               '--output_stride', '16', '--is_stride_conv', 'True', '--is_trans_conv', 'True'])
    assert dicearr.shape == (6, 5) and np.isfinite(dicearr[~np.isnan(dicearr)]).all()
