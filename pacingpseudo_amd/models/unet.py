"""U-Net backbone of PacingPseudo for MI355X.

Drop-in for the reference's ``models/unet.py`` (constructor signature, sub-module names and therefore
``state_dict`` keys, ``forward(x) -> end_points dict``; reference: models/unet.py:10-98).  The modules below
only *hold* parameters (stock ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects built in the reference's construction
order, so ``torch.manual_seed(s)`` yields the same initial weights as the reference): none of their
``forward`` methods is ever invoked.  All arithmetic runs in hand-written HIP kernels driven by
``pacingpseudo_amd.engine.StepEngine`` on NHWC activations.

Both variants of the reference are supported: max-pool + bilinear up-sampling (``is_stride_conv == is_trans_conv == False``,
the default and the one every reference configuration uses) and strided convolution + ``ConvTranspose2d``
(``--is_stride_conv / --is_trans_conv``, models/unet.py:100-152; built for correctness on top of the same kernels, see
csrc/pp_spatial.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class ConvLayer(nn.Module):
    """conv3x3 -> BatchNorm2d -> LeakyReLU(0.01) parameter holder (reference: models/unet.py:178-193)."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, dilation=1,
                 norm_op=nn.BatchNorm2d, nonlin_op=nn.LeakyReLU, negative_slop=1e-2):
        super().__init__()
        if kernel_size != 3 or stride not in (1, 2) or padding != dilation or (stride == 2 and dilation != 1):
            raise NotImplementedError('HIP path implements 3x3 convolutions with padding == dilation, stride 1 or (dilation 1) 2')
        if norm_op is not nn.BatchNorm2d or nonlin_op is not nn.LeakyReLU:
            raise NotImplementedError('HIP path implements BatchNorm2d + LeakyReLU blocks')
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, dilation)
        self.norm_op = norm_op(out_ch)
        self.nonlin_op = nonlin_op(negative_slop)
        self.dilation = dilation
        self.stride = stride

    def forward(self, x):
        raise RuntimeError('ConvLayer holds parameters only; run the model through UNet / ConsistencyRegulr')


class DoubleConv(nn.Module):
    """Two ConvLayers (reference: models/unet.py:154-176)."""

    def __init__(self, in_ch, out_ch, ks1=3, stride1=1, padding1=1, dilation1=1,
                 ks2=3, stride2=1, padding2=1, dilation2=1):
        super().__init__()
        self.conv_layer1 = ConvLayer(in_ch, out_ch, ks1, stride1, padding1, dilation1)
        self.conv_layer2 = ConvLayer(out_ch, out_ch, ks2, stride2, padding2, dilation2)

    def forward(self, x):
        raise RuntimeError('DoubleConv holds parameters only')


class EncBlock(nn.Module):
    """[MaxPool2d(2,2)] + DoubleConv, or a DoubleConv whose first convolution has stride 2 (reference: models/unet.py:100-127)."""

    def __init__(self, in_ch, out_ch, do_subsamp=True, is_stride_conv=False, dilation=1):
        super().__init__()
        self.pooling = nn.MaxPool2d(2, 2) if (do_subsamp and not is_stride_conv) else None
        stride1 = 2 if (do_subsamp and is_stride_conv) else 1
        self.conv_block = DoubleConv(in_ch, out_ch, 3, stride1, dilation, dilation, 3, 1, dilation, dilation)
        self.dilation = dilation
        self.stride = stride1

    def forward(self, x):
        raise RuntimeError('EncBlock holds parameters only')


class DecBlock(nn.Module):
    """up-sampling (bilinear with align_corners=True, or ConvTranspose2d(lower, skip, k, k, bias=False)) + concat with the skip +
    DoubleConv (reference: models/unet.py:129-152).  `up_ch`: channels of the up-sampled tensor in the concatenation."""

    def __init__(self, lower_ch, skip_ch, out_ch, trans_ks=2, trans_stride=2, is_trans_conv=False):
        super().__init__()
        self.trans = bool(is_trans_conv)
        if is_trans_conv:
            if trans_ks != trans_stride or trans_ks not in (1, 2):
                raise NotImplementedError('HIP path implements ConvTranspose2d with kernel == stride in (1, 2)')
            self.up_samp = nn.ConvTranspose2d(lower_ch, skip_ch, trans_ks, trans_stride, bias=False)   # unet.py:140
            self.conv_block = DoubleConv(2 * skip_ch, out_ch)
            self.up_ch = skip_ch
        else:
            self.up_samp = nn.Upsample(scale_factor=trans_stride, mode='bilinear', align_corners=True)
            self.conv_block = DoubleConv(lower_ch + skip_ch, skip_ch)
            self.up_ch = lower_ch
        self.scale = trans_stride
        self.lower_ch, self.skip_ch = lower_ch, skip_ch
        self.identity_up = (not is_trans_conv) and trans_stride == 1     # bilinear x1 with align_corners is an exact identity

    def forward(self, x, skip):
        raise RuntimeError('DecBlock holds parameters only')


class UNet(nn.Module):
    """Six encoder stages, five decoder stages, 1x1 head (reference: models/unet.py:10-98)."""

    def __init__(self, input_ch=1, init_ch=32, max_ch=512, num_classes=4, output_stride=32,
                 is_stride_conv=False, is_trans_conv=False, elab_end_points=False):
        super().__init__()
        assert is_trans_conv == is_stride_conv, \
            "Only combo of stride_conv and trans_conv or maxpool and upsample is allowed."
        assert output_stride in [8, 16, 32]
        self.elab_end_points = elab_end_points
        self.end_points = dict()          # ONE dict, updated in place by every forward (reference: unet.py:23)
        self.input_ch, self.num_classes, self.output_stride = input_ch, num_classes, output_stride
        ch = [min(max_ch, 2 ** k * init_ch) for k in range(6)]
        self.ch_ls = ch
        sc = is_stride_conv
        self.enc_block1 = EncBlock(input_ch, ch[0], do_subsamp=False, is_stride_conv=sc)
        self.enc_block2 = EncBlock(ch[0], ch[1], do_subsamp=True, is_stride_conv=sc)
        self.enc_block3 = EncBlock(ch[1], ch[2], do_subsamp=True, is_stride_conv=sc)
        self.enc_block4 = EncBlock(ch[2], ch[3], do_subsamp=True, is_stride_conv=sc)
        if output_stride == 32:
            sub5, dil5, sub6, dil6, up5, up4 = True, 1, True, 1, 2, 2
        elif output_stride == 16:
            sub5, dil5, sub6, dil6, up5, up4 = True, 1, False, 2, 1, 2
        else:
            sub5, dil5, sub6, dil6, up5, up4 = False, 2, False, 4, 1, 1
        self.enc_block5 = EncBlock(ch[3], ch[4], do_subsamp=sub5, is_stride_conv=sc, dilation=dil5)
        self.enc_block6 = EncBlock(ch[4], ch[5], do_subsamp=sub6, is_stride_conv=sc, dilation=dil6)
        self.dec_block5 = DecBlock(ch[5], ch[4], ch[4], up5, up5, is_trans_conv=is_trans_conv)
        self.dec_block4 = DecBlock(ch[4], ch[3], ch[3], up4, up4, is_trans_conv=is_trans_conv)
        self.dec_block3 = DecBlock(ch[3], ch[2], ch[2], is_trans_conv=is_trans_conv)
        self.dec_block2 = DecBlock(ch[2], ch[1], ch[1], is_trans_conv=is_trans_conv)
        self.dec_block1 = DecBlock(ch[1], ch[0], ch[0], is_trans_conv=is_trans_conv)
        self.final_conv = nn.Conv2d(ch[0], num_classes, 1, 1)
        self._engine = None               # stand-alone (inference) engine, created lazily
        # a loaded checkpoint (inference.py:138-146 through load_backbone) changes every weight: the stand-alone engine's
        # forward-only plans must re-pack their kernel-side layouts
        self.register_load_state_dict_post_hook(
            lambda module, incompatible_keys: module._engine.invalidate_packed() if module._engine is not None else None)

    # ---- helpers used by the engine -------------------------------------------------------------
    def enc_blocks(self):
        return [getattr(self, f'enc_block{k}') for k in range(1, 7)]

    def dec_blocks(self):
        return {k: getattr(self, f'dec_block{k}') for k in (5, 4, 3, 2, 1)}

    # ---- flat parameter / gradient slabs of the stand-alone (fully-supervised) backbone ------------------------
    def _ensure_flat(self):
        """Parameters of a bare UNet that is trained on its own (upper_bound_chaos.py:116-131) live in one FlatSlab,
        like ConsistencyRegulr's, so FusedAdam / FusedSGD update them in one launch."""
        from ..flat import FlatSlab
        flat = getattr(self, '_flat', None)
        w = self.final_conv.weight
        if flat is None or not flat.owns(w):
            flat = getattr(w, '_pp_flat', None)
            if flat is None or not flat.owns(w):          # not inside a ConsistencyRegulr slab either
                flat = FlatSlab([('backbone', [p for p in self.parameters() if p.requires_grad])])
            self._flat = flat
        return flat

    def forward(self, x):
        """Forward of the bare backbone: ``inference.py:104,159`` (no_grad) and the fully-supervised trainer
        ``upper_bound_chaos.py:156-171``, which back-propagates CE + Dice through ``end_points['segmentation/logits']``.
        With gradients enabled the logits carry ONE autograd node whose backward runs the engine's hand-derived
        backward plan into the flat gradient slab (gradients are overwritten per backward, not accumulated)."""
        from ..engine import StepEngine
        if self._engine is None:
            self._engine = StepEngine(self, None, None)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if not self.final_conv.weight.is_cuda:
                raise RuntimeError('pacingpseudo_amd runs on the GPU only: call model.cuda() first')
            self._ensure_flat()
            logits = _UNetFunction.apply(self, x, self.final_conv.weight)
            ep = dict(self._last_end_points)
            ep['segmentation/logits'] = logits
        else:
            ep = self._engine.infer_end_points(x, training=self.training)
        if not self.elab_end_points:
            ep = {'segmentation/logits': ep['segmentation/logits']}
        self.end_points.update(ep)
        return self.end_points


class _UNetFunction(torch.autograd.Function):
    """logits = UNet(x) with the engine's backward plan behind it."""

    @staticmethod
    def forward(ctx, net, x, anchor):
        ep = net._engine.unet_forward_train(x)
        ctx.net = net
        ctx.state = net._engine.last
        net._last_end_points = {k: v for k, v in ep.items() if k != 'segmentation/logits'}
        return ep['segmentation/logits']

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        flat = net._ensure_flat()
        net._engine.unet_backward(g, flat.grad_views, ctx.state)
        net._engine.unscale_grads(ctx.state, [flat.segment('backbone', 'grads')], flat)
        flat.publish_grads(['backbone'])
        return None, None, None
