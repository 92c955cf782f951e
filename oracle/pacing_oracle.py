"""CPU oracle for the PacingPseudo training step.  TEST INFRASTRUCTURE ONLY.

This file is a CPU (PyTorch fp32) restatement of the reference hot path.  It is
the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``pacingpseudo_amd/`` imports it, and the product path raises when the HIP
library is missing instead of falling back to this code.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md §4), so the pin is a set of vectors captured by importing the
reference itself on CPU in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks this restatement against every one.

Style: purely functional over a flat ``{state_dict key: tensor}`` mapping (the
reference is written as nn.Module classes).  Each function cites the reference
file:line it follows (paths relative to the upstream repository root).

The conv / batch-norm / pooling / interpolation arithmetic itself lives in
PyTorch (`aten`), a third-party dependency of the reference (pinned there as
torch 1.7.1, README.md:42); the oracle dispatches the same aten ops the
reference's call sites dispatch (models/unet.py:60,109,144,188-190,
models/aux_path_memory.py:22-33,52,75, losses/losses.py:16-17,33,43,54).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LRELU_SLOPE = 1e-2      # models/unet.py:167 (negative_slop=1e-2)
BN_EPS = 1e-5           # nn.BatchNorm2d default, models/unet.py:189
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# Configuration helpers
# --------------------------------------------------------------------------
# flag namespace and synthetic batch recipe live in the package (bench.py must not need oracle/ to build its inputs)
from pacingpseudo_amd.data import default_args, full_flags, synthetic_batch  # noqa: E402,F401


def stage_plan(args) -> dict:
    """Channel / pooling / dilation / up-sampling plan of models/unet.py:27-58."""
    ch = [min(args.max_ch, (2 ** k) * args.init_ch) for k in range(6)]
    os_ = args.output_stride
    assert os_ in (8, 16, 32)                      # models/unet.py:33
    if os_ == 32:
        pool5, dil5, pool6, dil6, up5, up4 = True, 1, True, 1, 2, 2
    elif os_ == 16:
        pool5, dil5, pool6, dil6, up5, up4 = True, 1, False, 2, 1, 2
    else:
        pool5, dil5, pool6, dil6, up5, up4 = False, 2, False, 4, 1, 1
    enc = [
        dict(cin=args.input_ch, cout=ch[0], pool=False, dil=1),
        dict(cin=ch[0], cout=ch[1], pool=True, dil=1),
        dict(cin=ch[1], cout=ch[2], pool=True, dil=1),
        dict(cin=ch[2], cout=ch[3], pool=True, dil=1),
        dict(cin=ch[3], cout=ch[4], pool=pool5, dil=dil5),
        dict(cin=ch[4], cout=ch[5], pool=pool6, dil=dil6),
    ]
    # --is_stride_conv / --is_trans_conv (models/unet.py:100-152): the sub-sampling stages use a stride-2 first convolution
    # instead of MaxPool2d, the decoder ConvTranspose2d(lower, skip, k, k, bias=False) instead of bilinear up-sampling
    sc = bool(getattr(args, 'is_stride_conv', False))
    assert sc == bool(getattr(args, 'is_trans_conv', False))          # models/unet.py:25
    for e in enc:
        e['stride'] = 2 if (sc and e['pool']) else 1
        if sc:
            e['pool'] = False
    # DecBlock(lower_ch, skip_ch, out_ch): conv in = lower+skip, out = skip  (models/unet.py:145)
    dec = {
        5: dict(lower=ch[5], skip=ch[4], up=up5),
        4: dict(lower=ch[4], skip=ch[3], up=up4),
        3: dict(lower=ch[3], skip=ch[2], up=2),
        2: dict(lower=ch[2], skip=ch[1], up=2),
        1: dict(lower=ch[1], skip=ch[0], up=2),
    }
    for d in dec.values():
        d['trans'] = sc
        d['cat_in'] = 2 * d['skip'] if sc else d['lower'] + d['skip']
    return dict(ch=ch, enc=enc, dec=dec)


def conv_layer_prefixes(args) -> List[str]:
    """state_dict prefixes of the 22 backbone ConvLayers, in construction order."""
    out = []
    for k in range(1, 7):
        for j in (1, 2):
            out.append(f'backbone.enc_block{k}.conv_block.conv_layer{j}')
    for k in (5, 4, 3, 2, 1):
        for j in (1, 2):
            out.append(f'backbone.dec_block{k}.conv_block.conv_layer{j}')
    return out


def init_state(args, seed: int = 1) -> Dict[str, Tensor]:
    """Fresh state_dict with PyTorch's default Conv2d/BatchNorm2d initialisation.

    Layout = the 165-entry state_dict of the reference model
    (train_chaos.py:188-213; SURVEY.md §5 checkpoint row).  Values are drawn with
    the oracle's own generator, they are NOT meant to reproduce the reference's
    RNG stream (golden fixtures carry explicit weights)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(prefix, cin, cout, k, bias=True):
        fan_in = cin * k * k
        bound = 1.0 / math.sqrt(fan_in)       # kaiming_uniform_(a=sqrt(5)) == U(+-1/sqrt(fan_in))
        sd[prefix + '.weight'] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        if bias:
            sd[prefix + '.bias'] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def bn(prefix, c):
        sd[prefix + '.weight'] = torch.ones(c)
        sd[prefix + '.bias'] = torch.zeros(c)
        sd[prefix + '.running_mean'] = torch.zeros(c)
        sd[prefix + '.running_var'] = torch.ones(c)
        sd[prefix + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)

    plan = stage_plan(args)
    for k, e in enumerate(plan['enc'], start=1):
        p = f'backbone.enc_block{k}.conv_block'
        conv(p + '.conv_layer1.conv', e['cin'], e['cout'], 3); bn(p + '.conv_layer1.norm_op', e['cout'])
        conv(p + '.conv_layer2.conv', e['cout'], e['cout'], 3); bn(p + '.conv_layer2.norm_op', e['cout'])
    for k in (5, 4, 3, 2, 1):
        d = plan['dec'][k]
        if d['trans']:          # nn.ConvTranspose2d weight [lower][skip][k][k], registered before the conv block (unet.py:140-141)
            kk = d['up']
            bound = 1.0 / math.sqrt(d['skip'] * kk * kk)
            sd[f'backbone.dec_block{k}.up_samp.weight'] = (torch.rand(d['lower'], d['skip'], kk, kk, generator=g) * 2 - 1) * bound
        p = f'backbone.dec_block{k}.conv_block'
        conv(p + '.conv_layer1.conv', d['cat_in'], d['skip'], 3); bn(p + '.conv_layer1.norm_op', d['skip'])
        conv(p + '.conv_layer2.conv', d['skip'], d['skip'], 3); bn(p + '.conv_layer2.norm_op', d['skip'])
    conv('backbone.final_conv', plan['ch'][0], args.num_classes, 1)
    conv('aux_path.layer_bottleneck.1', sum(args.feat_ch), args.hid_ch, 3)
    bn('aux_path.layer_bottleneck.2', args.hid_ch)
    conv('aux_path.fc_cls.1', args.hid_ch, args.num_classes, 1, bias=False)
    sd['aux_path.memory_bank'] = torch.zeros(args.num_classes, args.hid_ch, 1, 1)
    return sd


def trainable_keys(sd: Dict[str, Tensor]) -> List[str]:
    """Keys Adam sees with requires_grad=True (everything but BN buffers and the bank)."""
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked') or k == 'aux_path.memory_bank')]


# --------------------------------------------------------------------------
# Network
# --------------------------------------------------------------------------
def _bn(sd, prefix: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d forward incl. running-stat update (models/unet.py:189,193)."""
    if training:
        sd[prefix + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'],
                        training, BN_MOMENTUM, BN_EPS)


TAP = None       # debug aid: set to {} to record (z, y) of every ConvLayer call, in call order
MASKS = None     # test aid: {layer prefix: [bool mask per call]} -- see leaky_relu_choice()
MASK_STATS = []  # (prefix, #elements whose forced branch differs from pre > 0, max |pre| among those)
_CALLS: Dict[str, int] = {}


class _LeakyChoice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, mask):
        ctx.save_for_backward(mask)
        return torch.where(mask, pre, pre * LRELU_SLOPE)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * LRELU_SLOPE), None


def leaky_relu_choice(pre: Tensor, prefix: str) -> Tensor:
    """LeakyReLU(0.01).  LeakyReLU has no derivative at 0, and fp32 rounding decides the branch of an activation
    whose pre-activation is within ~1e-6 of 0.  When a test provides MASKS (the branch the device kernel took for
    every element), that choice is used instead of `pre > 0`, so gradients can be compared tightly; MASK_STATS
    records how many elements that changed and how close to 0 they were (the test bounds both)."""
    idx = _CALLS.get(prefix, 0)
    _CALLS[prefix] = idx + 1
    if MASKS is None or prefix not in MASKS:
        return F.leaky_relu(pre, LRELU_SLOPE)
    mask = MASKS[prefix][idx]
    diff = mask != (pre.detach() > 0)
    n = int(diff.sum())
    MASK_STATS.append((prefix, n, float(pre.detach().abs()[diff].max()) if n else 0.0))
    return _LeakyChoice.apply(pre, mask)


POOLS = None     # test aid: {pooling key: [window-position index (N,C,H/2,W/2) per call]} -- see max_pool_choice()
POOL_STATS = []  # (key, #windows whose forced winner differs from the arg-max, largest value gap among those)


class _PoolChoice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        N, C, H, W = x.shape
        win = x.view(N, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H // 2, W // 2, 4)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return win.gather(-1, idx.unsqueeze(-1)).squeeze(-1)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        win = torch.zeros(N, C, H // 2, W // 2, 4, dtype=g.dtype)
        win.scatter_(-1, idx.unsqueeze(-1), g.unsqueeze(-1))
        return win.view(N, C, H // 2, W // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H, W), None


def max_pool_choice(x: Tensor, key: str) -> Tensor:
    """MaxPool2d(2,2).  Like the LeakyReLU kink, the winner of a 2x2 window whose two largest values agree to fp32
    rounding is decided by the last bit; a test may pass POOLS (the winner the device kernel picked) so that the
    gradient is routed to the same pixel.  POOL_STATS records how many windows that changed and by what margin."""
    idx_call = _CALLS.get(key, 0)
    _CALLS[key] = idx_call + 1
    if POOLS is None or key not in POOLS:
        return F.max_pool2d(x, 2, 2)
    idx = POOLS[key][idx_call]
    out = _PoolChoice.apply(x, idx)
    nat = F.max_pool2d(x.detach(), 2, 2)
    diff = out.detach() != nat
    n = int(diff.sum())
    POOL_STATS.append((key, n, float((nat - out.detach())[diff].max()) if n else 0.0))
    return out


def conv_layer(sd, prefix: str, x: Tensor, dil: int, training: bool, stride: int = 1) -> Tensor:
    """ConvLayer.forward: conv3x3(pad=dil) -> BN -> LeakyReLU(0.01)  (models/unet.py:188-193)."""
    w = sd[prefix + '.conv.weight']
    if CONV_OPERAND_ROUND is not None:         # mixed-precision check: the device rounds both conv operands to fp16
        x, w = CONV_OPERAND_ROUND(x, prefix), CONV_OPERAND_ROUND(w, prefix)
    z = F.conv2d(x, w, sd[prefix + '.conv.bias'], stride, dil, dil)
    y = leaky_relu_choice(_bn(sd, prefix + '.norm_op', z, training), prefix)
    if TAP is not None and y.requires_grad:
        z.retain_grad(); y.retain_grad()
        TAP.setdefault(prefix, []).append((z, y))
    return y


# hook for tests of the fp16-operand mode (`--precision fp16`): callable(tensor, layer prefix) -> tensor, or None
CONV_OPERAND_ROUND = None


def double_conv(sd, prefix: str, x: Tensor, dil: int, training: bool, stride1: int = 1) -> Tensor:
    """DoubleConv.forward (models/unet.py:175-176); stride1 = 2 for the sub-sampling stages under --is_stride_conv (:113-116)."""
    return conv_layer(sd, prefix + '.conv_layer2', conv_layer(sd, prefix + '.conv_layer1', x, dil, training, stride1),
                      dil, training)


def unet_forward(sd, x: Tensor, args, training: bool) -> Dict[str, Tensor]:
    """UNet.forward (models/unet.py:62-98), max-pool / bilinear variant only."""
    plan = stage_plan(args)
    enc = []
    h = x
    for k, e in enumerate(plan['enc'], start=1):
        if e['pool']:
            h = max_pool_choice(h, f'backbone.enc_block{k}.pooling')    # models/unet.py:109,124-125
        h = double_conv(sd, f'backbone.enc_block{k}.conv_block', h, e['dil'], training, e['stride'])
        enc.append(h)
    d = enc[5]
    decs = {}
    for k in (5, 4, 3, 2, 1):
        up = plan['dec'][k]['up']
        # nn.Upsample(scale_factor, bilinear, align_corners=True)  (models/unet.py:144,149-150)
        if plan['dec'][k]['trans']:          # nn.ConvTranspose2d(lower, skip, up, up, bias=False)  (models/unet.py:140,149)
            u = F.conv_transpose2d(d, sd[f'backbone.dec_block{k}.up_samp.weight'], None, up)
        else:
            u = F.interpolate(d, scale_factor=up, mode='bilinear', align_corners=True)
        d = double_conv(sd, f'backbone.dec_block{k}.conv_block', torch.cat((u, enc[k - 1]), 1), 1, training)
        decs[k] = d
    logits = F.conv2d(d, sd['backbone.final_conv.weight'], sd['backbone.final_conv.bias'])   # :60,75
    ep = {f'encoder/stage{k}': enc[k - 1] for k in range(1, 7)}
    ep.update({f'decoder/stage{k}': decs[k] for k in (5, 4, 3, 2, 1)})
    ep['segmentation/logits'] = logits
    return ep


def ramp_up_mo(step, max_step, base_mo=0.9, gamma=0.9) -> float:
    """models/aux_path_memory.py:118-120."""
    return (1 - step / max_step) ** gamma * base_mo


@torch.no_grad()
def memory_update(bank: Tensor, aux_features: Tensor, scribble: Tensor, step, args) -> None:
    """AuxPath.memory_update (models/aux_path_memory.py:68-116), in place on ``bank`` (K,hid,1,1).

    Only batch sample 0 contributes: the reference returns from inside the
    per-sample loop (models/aux_path_memory.py:116)."""
    K = args.num_classes
    h, w = scribble.shape[-2:]
    feat = F.interpolate(aux_features[:1], size=(h, w), mode='bilinear', align_corners=True)   # :75
    emb_all = feat[0].permute(1, 2, 0).reshape(h * w, -1)                # (h*w, hid)
    scb = scribble[0].permute(1, 2, 0).reshape(h * w, -1)                # (h*w, K+1)
    for c in range(K):
        mask = scb[:, c] == 1                                            # :83
        if not bool(mask.sum()):
            continue
        emb = emb_all[mask].clone()                                      # (N_c, hid)
        row = bank[c, :, 0, 0]                                           # view of the bank row
        if int((row == 0).sum()) == args.hid_ch:                         # :92 first visit -> plain mean
            new = emb.mean(0)
        else:
            if args.ensemble_mode == 'mean':
                upd = emb.mean(0)
            else:                                                        # cosine_similarity :101-110
                emb = emb / (emb.pow(2).sum(1, keepdim=True).sqrt() + 1e-8)
                row /= (row.pow(2).sum().sqrt() + 1e-8)                  # in place on the bank row (:106)
                cos = (emb * row[None]).sum(1, keepdim=True)
                wts = (1 - cos) / ((1 - cos).sum() + 1e-8)
                upd = (emb * wts).sum(0)
            m = ramp_up_mo(step, args.epoch, args.update_momentum)       # max_step = args.epoch (:208)
            new = (1 - m) * row + m * upd
        bank[c, :, 0, 0] = new


DROP_MASKS = None   # test aid: {'input' | 'features' | 'bank': (N,C) multipliers (0 or 1/(1-p))} replacing the RNG draw


def _dropout2d(x: Tensor, p: float, training: bool, key: str) -> Tensor:
    """nn.Dropout2d (models/aux_path_memory.py:22,31): whole channels of a sample are zeroed with probability p,
    survivors scaled by 1/(1-p); identity in eval mode.  A test may force the mask the device drew."""
    if DROP_MASKS is not None and key in DROP_MASKS and training and p > 0:
        return x * DROP_MASKS[key][:, :, None, None]
    return F.dropout2d(x, p, training)


def aux_forward(sd, end_points, scribble: Tensor, step, args, training: bool) -> Dict[str, Tensor]:
    """AuxPath.forward (models/aux_path_memory.py:46-66)."""
    feat = torch.cat([end_points[s] for s in args.feat_stage], 1)
    feat = _dropout2d(feat, args.aux_drop_prob, training, 'input')
    z = F.conv2d(feat, sd['aux_path.layer_bottleneck.1.weight'], sd['aux_path.layer_bottleneck.1.bias'], 1, 1)
    aux_features = leaky_relu_choice(_bn(sd, 'aux_path.layer_bottleneck.2', z, training), 'aux_path.layer_bottleneck')
    lo = F.conv2d(_dropout2d(aux_features, args.aux_drop_prob, training, 'features'), sd['aux_path.fc_cls.1.weight'])
    logits_aux = F.interpolate(lo, size=scribble.shape[-2:], mode='bilinear', align_corners=True)
    out = {'logits_aux_cls': logits_aux, 'aux_targets': scribble.argmax(1).long(), 'aux_features': aux_features}
    if args.do_memory:
        memory_update(sd['aux_path.memory_bank'], aux_features.detach(), scribble, step, args)
        out['logits_memory'] = F.conv2d(_dropout2d(sd['aux_path.memory_bank'], args.aux_drop_prob, training, 'bank'),
                                        sd['aux_path.fc_cls.1.weight'])
        out['memory_target'] = torch.arange(args.num_classes, dtype=torch.long)
    return out


# --------------------------------------------------------------------------
# Losses (losses/losses.py) -- written out as explicit arithmetic
# --------------------------------------------------------------------------
def partial_cross_entropy_loss(logits: Tensor, target: Tensor, ignore_index: int) -> Tensor:
    """losses/losses.py:35-43 == mean over non-ignored pixels of -log_softmax[target]; 0 valid -> NaN."""
    lsm = torch.log_softmax(logits, 1)
    keep = target != ignore_index
    picked = lsm.gather(1, target.clamp(max=logits.shape[1] - 1).unsqueeze(1)).squeeze(1)
    return -(picked * keep).sum() / keep.sum()


def cross_entropy_loss(logits: Tensor, target: Tensor) -> Tensor:
    """losses/losses.py:26-33 (plain mean CE on (N,C))."""
    lsm = torch.log_softmax(logits, 1)
    return -lsm.gather(1, target.unsqueeze(1)).mean()


def _masked_mean(per_elem: Tensor, valid_mask: Optional[Tensor]) -> Tensor:
    if valid_mask is None:
        return per_elem.mean()
    return (per_elem * valid_mask).sum() / max(valid_mask.sum(), 1e-8)


def entropy_minimization_loss(logits: Tensor, valid_mask=None) -> Tensor:
    """losses/losses.py:9-24."""
    return _masked_mean(-torch.softmax(logits, 1) * torch.log_softmax(logits, 1), valid_mask)


def soft_label_cross_entropy_loss(logits: Tensor, target_prob: Tensor, valid_mask=None) -> Tensor:
    """losses/losses.py:45-62."""
    return _masked_mean(-target_prob * torch.log_softmax(logits, 1), valid_mask)


def l1_loss(p: Tensor, q: Tensor, valid_mask=None) -> Tensor:
    """losses/losses.py:64-79."""
    return _masked_mean((p - q).abs().sum(1, keepdim=True), valid_mask)


def l2_loss(p: Tensor, q: Tensor, valid_mask=None) -> Tensor:
    """losses/losses.py:81-96."""
    return _masked_mean((p - q).pow(2).sum(1, keepdim=True), valid_mask)


def kl_loss(logits: Tensor, target_logits: Tensor, valid_mask=None) -> Tensor:
    """losses/losses.py:98-116: KL(target || input) element-wise, log_target form."""
    li, lt = torch.log_softmax(logits, 1), torch.log_softmax(target_logits, 1)
    return _masked_mean(lt.exp() * (lt - li), valid_mask)


def dice_loss_fn(logits: Tensor, target: Tensor) -> Tensor:
    """losses/losses.py:147-162: -mean over (n, c) of 2 sum(p t) / (sum p + sum t + 1e-5), p = softmax(logits)."""
    p = torch.softmax(logits, 1).flatten(2)
    t = target.flatten(2)
    return -((2 * (p * t).sum(2)) / (p.sum(2) + t.sum(2) + 1e-5)).mean()


def upper_bound_losses(sd, image: Tensor, label: Tensor, args, training: bool, loss_dice: bool = True):
    """Loss assembly of the fully-supervised trainer (upper_bound_chaos.py:156-168): bare UNet, partial CE against
    argmax(label) + soft Dice loss.  `sd` uses the `backbone.` key prefix of the composite model."""
    logits = unet_forward(sd, image, args, training)['segmentation/logits']
    loss_ce = partial_cross_entropy_loss(logits, label.argmax(1).long(), args.ignored_index)
    out = {'segmentation/logits': logits, 'loss_ce': loss_ce}
    if loss_dice:
        out['loss_dice'] = dice_loss_fn(logits, label)
    return out


# --------------------------------------------------------------------------
# Composite forward (models/consistency_reglur_memory.py:24-102)
# --------------------------------------------------------------------------
def consistency_forward(sd, batch: Dict[str, Tensor], mode, step, args, training: bool) -> Dict[str, Tensor]:
    assert mode in ('train', 'val', None)
    out: Dict[str, Tensor] = {}
    end_points = unet_forward(sd, batch['image'], args, training)                      # :29
    logits_weak = end_points['segmentation/logits']
    scb_target = batch['scribble'].argmax(1).long()                                    # :31
    out['segmentation/logits'] = logits_weak
    out['loss_pce'] = partial_cross_entropy_loss(logits_weak, scb_target, args.ignored_index)
    valid_mask = batch.get('valid_mask')
    if mode == 'train' and args.do_loss_ent:                                           # :40
        out['loss_ent'] = entropy_minimization_loss(logits_weak, valid_mask)
    if mode == 'train' and args.do_decoder_consistency:                                # :47
        # The reference's UNet keeps ONE end_points dict and updates it in place
        # (models/unet.py:23,82-98), so after this call `end_points` holds the STRONG view.
        end_points = unet_forward(sd, batch['image_strong'], args, training)
        logits_strong = end_points['segmentation/logits']
        prob_weak = torch.softmax(logits_weak, 1)
        if args.detach_weak_cr:
            prob_weak = prob_weak.detach()
        v = args.loss_cr_variants
        if v == 'ce_loss':
            loss_cr = soft_label_cross_entropy_loss(logits_strong, prob_weak, valid_mask)
        elif v == 'l1_loss':
            loss_cr = l1_loss(torch.softmax(logits_strong, 1), prob_weak, valid_mask)
        elif v == 'l2_loss':
            loss_cr = l2_loss(torch.softmax(logits_strong, 1), prob_weak, valid_mask)
        elif v == 'kl_loss':
            loss_cr = kl_loss(logits_strong, logits_weak, valid_mask)
        else:
            raise ValueError('The loss is not implemented.')                           # :65
        out['loss_cr'] = loss_cr
        out['segmentation/logits_strong'] = logits_strong
    if mode == 'train' and args.do_aux_path:                                           # :73
        aux = aux_forward(sd, end_points, batch['scribble'], step, args, training)
        out['logits_aux_cls'] = aux['logits_aux_cls']
        out['loss_aux_cls'] = partial_cross_entropy_loss(aux['logits_aux_cls'], aux['aux_targets'],
                                                         args.ignored_index)
        out['_aux_features'] = aux['aux_features']
        if args.do_memory:
            out['loss_memory'] = cross_entropy_loss(aux['logits_memory'].squeeze(-1).squeeze(-1),
                                                    aux['memory_target'])
    return out


# --------------------------------------------------------------------------
# Scalar helpers (utils/utils.py, utils/metrics.py)
# --------------------------------------------------------------------------
def gaussian_ramp_up(t, base_value, max_t=80, scale=5.0) -> float:
    """utils/utils.py:53-65."""
    return base_value * math.exp(-scale * (1 - t / max_t)) if t < max_t else base_value


def lr_at(policy: str, step, num_steps, base_lr, gamma=0.9) -> float:
    """utils/utils.py:7-51 (value only; the reference also writes it into the optimizer)."""
    if policy == 'poly':
        return base_lr * (1 - step / num_steps) ** gamma
    if policy == 'cosine':
        return 0.5 * (1 + math.cos(step * math.pi / num_steps)) * base_lr
    if policy == 'linear':
        return (1 - step / num_steps) * base_lr
    raise ValueError('Unimplemented learning rate decay policy.')                      # train_chaos.py:260


def compute_dice(prob: np.ndarray, target: np.ndarray) -> list:
    """utils/metrics.py:7-34: per-class Dice of argmax(prob) vs one-hot target, NaN if both empty."""
    assert prob.shape == target.shape
    C = prob.shape[0]
    pred = np.argmax(prob, axis=0)
    out = []
    for c in range(C):
        p = (pred == c).astype(np.float64).reshape(-1)
        t = target[c].reshape(-1)
        if not p.any() and not t.any():
            out.append(np.nan)
        else:
            out.append(2 * np.sum(p * t) / (np.sum(p) + np.sum(t) + 1e-5))
    return out


def skeletonize_zhang(mask: np.ndarray) -> np.ndarray:
    """skimage.morphology.skeletonize (2-D) restated.  skimage is absent here and unpinned in the reference (imported
    at utils/utils_artificial_scribbles.py:3), so this is its published algorithm -- Zhang & Suen 1984 two-sub-iteration
    thinning, all deletions of a sub-iteration applied together -- written with numpy shifts; PARITY UNPINNED for this
    helper (no skimage output to compare with)."""
    sk = np.pad(mask.astype(bool), 1).astype(np.uint8)
    while True:
        changed = False
        for first in (True, False):
            p2, p3, p4 = sk[:-2, 1:-1], sk[:-2, 2:], sk[1:-1, 2:]
            p5, p6, p7 = sk[2:, 2:], sk[2:, 1:-1], sk[2:, :-2]
            p8, p9 = sk[1:-1, :-2], sk[:-2, :-2]
            ring = [p2, p3, p4, p5, p6, p7, p8, p9, p2]
            B = sum(ring[:8]).astype(np.int32)
            A = sum(((ring[i] == 0) & (ring[i + 1] == 1)).astype(np.int32) for i in range(8))
            c = (sk[1:-1, 1:-1] == 1) & (B >= 2) & (B <= 6) & (A == 1)
            if first:
                c &= (p2 * p4 * p6 == 0) & (p4 * p6 * p8 == 0)
            else:
                c &= (p2 * p4 * p8 == 0) & (p2 * p6 * p8 == 0)
            if c.any():
                sk[1:-1, 1:-1][c] = 0
                changed = True
        if not changed:
            return sk[1:-1, 1:-1].astype(bool)


def generate_scribble_fn(lab: np.ndarray, num_classes: int, ignored_index: int) -> np.ndarray:
    """utils/utils_artificial_scribbles.py:5-35."""
    from scipy import ndimage
    h, w = lab.shape
    lab_oh = np.zeros((num_classes, h, w))
    scb_oh = np.zeros_like(lab_oh)
    for c in range(num_classes):
        lab_oh[c][lab == c] = 1
        scb_oh[c] = skeletonize_zhang(lab_oh[c]) * lab_oh[c]
    scb_oh = np.concatenate([scb_oh, 1 - np.sum(scb_oh, axis=0, keepdims=True)], axis=0)
    scb_bg = scb_oh[0]
    if set(np.unique(np.argmax(scb_oh, axis=0))) == {0, ignored_index}:
        scb_bg = ndimage.binary_dilation(scb_oh[0], np.eye(3)[::-1], iterations=40, mask=lab_oh[0])
        scb_bg = skeletonize_zhang(scb_bg)
    scb_oh[0] = scb_bg
    return np.argmax(scb_oh, axis=0)


def compute_95hd(pred_hard: np.ndarray, label: np.ndarray, num_classes: int, spacing) -> list:
    """inference.py:217-237.  The reference delegates to ``medpy.metric.binary.hd95(result, reference, voxelspacing,
    connectivity=1)``; medpy is a third-party dependency that is absent here and unpinned in the reference (README.md
    lists it without a version), so its published algorithm (medpy 0.4.0, metric/binary.py: hd95 +
    __surface_distances) is restated on scipy.ndimage: border = mask XOR binary_erosion(mask, cross structure),
    dt = distance_transform_edt(~reference_border, sampling), sds = dt[result_border], both directions,
    numpy.percentile(hstack, 95)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    fp = generate_binary_structure(2, 1)
    out = []
    for c in range(num_classes):
        a, b = pred_hard == c, label == c
        if not a.any() or not b.any() or a.all() or b.all():
            out.append(np.nan)
            continue
        ab = a ^ binary_erosion(a, structure=fp, iterations=1)
        bb = b ^ binary_erosion(b, structure=fp, iterations=1)
        d1 = distance_transform_edt(~bb, sampling=spacing)[ab]
        d2 = distance_transform_edt(~ab, sampling=spacing)[bb]
        out.append(float(np.percentile(np.hstack((d1, d2)), 95)))
    return out


# --------------------------------------------------------------------------
# One training iteration (train_chaos.py:263-315) with Adam (train_chaos.py:219)
# --------------------------------------------------------------------------
def loss_weights(args, epoch) -> Dict[str, float]:
    """Weights the iteration body multiplies into each loss (train_chaos.py:273-310)."""
    w = {'loss_pce': 1.0}
    if args.do_loss_ent:
        w['loss_ent'] = (gaussian_ramp_up(epoch, args.loss_ent_weight, scale=args.ramp_up_scale)
                         if args.ramp_up_loss_ent else 1.0)
    if args.do_decoder_consistency:
        w['loss_cr'] = (gaussian_ramp_up(epoch, args.loss_cr_weight, scale=args.ramp_up_scale)
                        if args.ramp_up_loss_cr else 1.0)
    if args.do_aux_path:
        w['loss_aux_cls'] = args.loss_aux_weight
        if args.do_memory:
            w['loss_memory'] = args.loss_memory_weight
    return w


class AdamState:
    """torch.optim.Adam(params, lr, weight_decay=wd) restated (betas .9/.999, eps 1e-8, L2-coupled wd)."""

    def __init__(self):
        self.t = 0
        self.m: Dict[str, Tensor] = {}
        self.v: Dict[str, Tensor] = {}

    @torch.no_grad()
    def step(self, sd, grads: Dict[str, Optional[Tensor]], lr: float, wd: float, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k, g in grads.items():
            if g is None:                       # params that never got a grad are skipped by torch
                continue
            p = sd[k]
            g = g + wd * p
            if k not in self.m:
                self.m[k] = torch.zeros_like(p); self.v[k] = torch.zeros_like(p)
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(self.m[k], denom, value=-lr / bc1)


def train_step(sd, batch, epoch: int, args, training: bool, adam: Optional[AdamState] = None,
               lr: Optional[float] = None):
    """One iteration of train_chaos.py:263-315.  Returns (net_outputs, grads dict, total loss)."""
    keys = trainable_keys(sd)
    _CALLS.clear()
    del MASK_STATS[:]
    del POOL_STATS[:]
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    out = consistency_forward(sd, batch, 'train', epoch, args, training)
    w = loss_weights(args, epoch)
    total = sum(out[name] * wt for name, wt in w.items())
    total.backward()
    grads = {k: (sd[k].grad.detach().clone() if sd[k].grad is not None else None) for k in keys}
    for k in keys:
        sd[k].requires_grad_(False)
        sd[k].grad = None
    if adam is not None:
        adam.step(sd, grads, lr if lr is not None else lr_at(args.lr_decay, epoch, args.epoch, args.lr), args.wd)
    return {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in out.items()}, grads, float(total.detach())
