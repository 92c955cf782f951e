#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference ships no tests or golden vectors, so these captured
vectors are what pins the oracle (oracle/pacing_oracle.py) and, through it, the
HIP path.  Only DATA is written (inputs, weights, outputs, gradients); no
reference source text is stored.

What is imported from the reference: models/consistency_reglur_memory.py (which
pulls models/unet.py, models/aux_path_memory.py, losses/losses.py), utils/utils.py,
utils/metrics.py.  train_chaos.py itself cannot be imported (cv2 / skimage /
tensorboard are absent), so its 50-line iteration body (train_chaos.py:263-315)
is driven from here with the reference's own model, loss-weight helpers and
torch.optim.Adam.  One harness-side shim: ``torch.Tensor.cuda`` is made the
identity because AuxPath.__init__ calls ``.cuda()`` (aux_path_memory.py:44).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

REF = os.environ.get('PP_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self          # harness-side only
torch.nn.Module.cuda = lambda self, *a, **k: self

from models.consistency_reglur_memory import ConsistencyRegulr   # noqa: E402
from utils.utils import gaussian_ramp_up, poly_lr_decay          # noqa: E402
from utils.metrics import compute_dice                           # noqa: E402


def make_args(**over):
    a = dict(input_ch=1, init_ch=4, max_ch=32, num_classes=5, output_stride=8,
             is_stride_conv=False, is_trans_conv=False, elab_end_points=True,
             ignored_index=5, epoch=400, lr=1e-4, wd=3e-4,
             do_loss_ent=False, loss_ent_weight=1.0, ramp_up_loss_ent=True, ramp_up_scale=8.0,
             do_decoder_consistency=False, ramp_up_loss_cr=True, detach_weak_cr=False,
             loss_cr_variants='ce_loss', loss_cr_weight=1.0,
             do_aux_path=False, feat_stage=['encoder/stage6', 'encoder/stage5'], feat_ch=[32, 32],
             loss_aux_weight=0.01, hid_ch=8, aux_drop_prob=0.0,
             do_memory=False, loss_memory_weight=1.0, update_momentum=0.9,
             ensemble_mode='cosine_similarity')
    a.update(over)
    return SimpleNamespace(**a)


def build(args, seed=1):
    torch.manual_seed(seed)
    return ConsistencyRegulr(
        kwargs_unet=dict(input_ch=args.input_ch, init_ch=args.init_ch, max_ch=args.max_ch,
                         num_classes=args.num_classes, output_stride=args.output_stride,
                         is_stride_conv=args.is_stride_conv, is_trans_conv=args.is_trans_conv,
                         elab_end_points=args.elab_end_points),
        kwargs_aux_path=dict(num_classes=args.num_classes, feat_stage=args.feat_stage, feat_ch=args.feat_ch,
                             hid_ch=args.hid_ch, aux_drop_prob=args.aux_drop_prob, do_memory=args.do_memory,
                             max_step=args.epoch, update_momentum=args.update_momentum,
                             ensemble_mode=args.ensemble_mode),
        args_parser=args)


def make_batch(B, H, W, C=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 1, H, W, generator=g)
    a = torch.rand(B, 1, 1, 1, generator=g) * 1.6 + 0.2
    b = torch.rand(B, 1, 1, 1, generator=g) * 1.6 - 0.8
    image_strong = image * a + b
    coarse = torch.randint(0, C, (B, 1, H // 8, W // 8), generator=g).float()
    label = torch.nn.functional.interpolate(coarse, size=(H, W), mode='nearest').long().squeeze(1)
    kept = torch.rand(B, H, W, generator=g) < 0.06
    scb = torch.where(kept, label, torch.full_like(label, C))
    scb[0][scb[0] == C - 1] = C            # class C-1 has no scribble in sample 0 (the `continue` branch)
    scribble = torch.nn.functional.one_hot(scb, C + 1).permute(0, 3, 1, 2).float().contiguous()
    label_1h = torch.nn.functional.one_hot(label, C).permute(0, 3, 1, 2).float().contiguous()
    valid = torch.ones(B, 1, H, W)
    valid[:, :, :3, :] = 0                 # a cropped-out border, as RandomCrop would leave
    valid[:, :, :, -5:] = 0
    return dict(image=image, image_strong=image_strong, scribble=scribble, valid_mask=valid, label=label_1h)


def sd_np(model, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def iteration(model, optimizer, batch, args, epoch):
    """train_chaos.py:263-315 restated around the reference's own modules."""
    b = {k: v.clone() for k, v in batch.items() if k != 'label'}
    out = model(b, mode='train', step=epoch)
    rec = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
    loss = out['loss_pce']
    if args.do_loss_ent:
        le = out['loss_ent']
        if args.ramp_up_loss_ent:
            le = le * gaussian_ramp_up(t=epoch, base_value=args.loss_ent_weight, scale=args.ramp_up_scale)
        loss = loss + le
    if args.do_decoder_consistency:
        lc = out['loss_cr']
        if args.ramp_up_loss_cr:
            lc = lc * gaussian_ramp_up(t=epoch, base_value=args.loss_cr_weight, scale=args.ramp_up_scale)
        loss = loss + lc
    if args.do_aux_path:
        loss = loss + out['loss_aux_cls'] * args.loss_aux_weight
        if args.do_memory:
            loss = loss + out['loss_memory'] * args.loss_memory_weight
    optimizer.zero_grad()
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    optimizer.step()
    rec['total_loss'] = loss.detach().clone()
    return rec, grads


def dump(path, d):
    flat = {}
    for k, v in d.items():
        if v is None:
            continue
        flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(path, **flat)
    print(f'wrote {path}: {len(flat)} arrays, {os.path.getsize(path) / 1e6:.2f} MB')


SUBSET = ('backbone.final_conv.weight', 'backbone.final_conv.bias',
          'backbone.enc_block1.conv_block.conv_layer1.conv.weight',
          'backbone.enc_block6.conv_block.conv_layer2.norm_op.weight',
          'backbone.dec_block3.conv_block.conv_layer1.conv.weight',
          'backbone.dec_block5.up_samp.weight', 'backbone.dec_block4.up_samp.weight', 'backbone.dec_block1.up_samp.weight',
          'backbone.enc_block2.conv_block.conv_layer1.conv.weight', 'backbone.enc_block4.conv_block.conv_layer1.conv.weight',
          'aux_path.fc_cls.1.weight', 'aux_path.layer_bottleneck.1.bias')


def sequence(name, args, epochs, eval_after_first_epoch=True, B=2, H=64, W=64, slim=False):
    """Run len(epochs) iterations, one per listed epoch index, switching to model.eval() once the
    epoch index changes (train_chaos.py:370 is never undone)."""
    model = build(args)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    out = {}
    batch0 = None
    out.update({'init/' + k: v for k, v in sd_np(model, '').items()})
    prev_epoch = epochs[0]
    for i, ep in enumerate(epochs):
        if ep != prev_epoch and eval_after_first_epoch:
            model.eval()
        prev_epoch = ep
        opt, lr = poly_lr_decay(opt, ep, args.epoch, args.lr)
        batch = make_batch(B, H, W, args.num_classes, seed=100 + i)
        if batch0 is None:
            batch0 = batch
        for k, v in batch.items():
            out[f'step{i}/in/{k}'] = v.numpy()
        rec, grads = iteration(model, opt, batch, args, ep)
        out[f'step{i}/epoch'] = np.asarray(ep)
        out[f'step{i}/lr'] = np.asarray(lr)
        out[f'step{i}/bn_training'] = np.asarray(int(model.training))
        for k, v in rec.items():
            out[f'step{i}/out/{k}'] = v.numpy()
        for k, g in grads.items():
            if g is not None and (not slim or k in SUBSET):
                out[f'step{i}/grad/{k}'] = g.numpy()
        post = sd_np(model, '')
        out.update({f'step{i}/post/' + k: v for k, v in post.items()
                    if not slim or k in SUBSET or k == 'aux_path.memory_bank'})
    # validation forward + Dice (train_chaos.py:370-392) on the first batch
    model.eval()
    with torch.no_grad():
        vo = model({k: v.clone() for k, v in batch0.items()}, mode='val')
    out['val/logits'] = vo['segmentation/logits'].numpy()
    out['val/loss_pce'] = vo['loss_pce'].numpy()
    sm = torch.softmax(vo['segmentation/logits'], 1).numpy()
    out['val/dice'] = np.asarray([compute_dice(sm[n], batch0['label'].numpy()[n]) for n in range(sm.shape[0])],
                                 dtype=np.float64)
    out['val/keys'] = np.asarray(sorted(vo.keys()))
    dump(os.path.join(HERE, name + '.npz'), out)


def main():
    torch.set_num_threads(4)
    full = dict(do_loss_ent=True, do_decoder_consistency=True, do_aux_path=True, do_memory=True)
    if sys.argv[1:] == ['strideconv']:       # round 3: only the --is_stride_conv / --is_trans_conv captures (the others are unchanged)
        sc = dict(is_stride_conv=True, is_trans_conv=True)
        # stride 8: ConvTranspose2d with kernel == stride == 1 in dec5 / dec4; one train-mode and one eval-mode BN iteration
        sequence('strideconv8', make_args(**full, **sc), epochs=[0, 1])
        # stride 16: both ConvTranspose2d kernel sizes next to the aux path and the memory bank
        sequence('strideconv16', make_args(**full, **sc, output_stride=16), epochs=[0], slim=True)
        # stride 32: five stride-2 convolutions, five 2x2 transposed convolutions (aux path rejected by the reference, see below)
        sequence('strideconv32', make_args(do_loss_ent=True, do_decoder_consistency=True, **sc, output_stride=32),
                 epochs=[0], slim=True)
        return
    # (A) full flags: two train-mode-BN iterations in epoch 0 (bank first-visit, then cosine update),
    #     then one eval-mode-BN iteration in epoch 1.
    sequence('full_seq', make_args(**full), epochs=[0, 0, 1])
    # (B) Control session (pCE only), one train-mode iteration.
    sequence('control_seq', make_args(), epochs=[0])
    # (C) consistency-loss variants + detached weak branch + 'mean' ensemble, late epoch (ramp-up == 1).
    sequence('variant_l1', make_args(**full, loss_cr_variants='l1_loss'), epochs=[100], slim=True)
    sequence('variant_l2', make_args(**full, loss_cr_variants='l2_loss', detach_weak_cr=True), epochs=[100], slim=True)
    sequence('variant_kl', make_args(**full, loss_cr_variants='kl_loss', ensemble_mode='mean'), epochs=[37, 37])
    # (D) other strides of the encoder (models/unet.py:34-53)
    sequence('stride16', make_args(**full, output_stride=16), epochs=[0], slim=True)
    # stride 32 + aux path is rejected by the reference itself (stage6/stage5 sizes differ,
    # aux_path_memory.py:49), so that stride is captured without the aux path.
    sequence('stride32', make_args(do_loss_ent=True, do_decoder_consistency=True, output_stride=32),
             epochs=[0], slim=True)


if __name__ == '__main__':
    main()
