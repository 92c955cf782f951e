"""Auxiliary path + class-prototype memory bank (drop-in for the reference's models/aux_path_memory.py).

``layer_bottleneck`` / ``fc_cls`` / ``memory_bank`` keep the reference's names and shapes (aux_path_memory.py:21-43) so
checkpoints interchange.  Inside ``ConsistencyRegulr`` the arithmetic (3x3 conv + BN + LeakyReLU, Dropout2d masks, 1x1
classifier, x8 bilinear up-sampling fused with partial CE, memory update of batch sample 0, bank classification) is part of
``pacingpseudo_amd.engine.StepEngine``'s fused step; ``AuxPath.forward(end_points, scribble, step)`` is the reference's
stand-alone call (aux_path_memory.py:46-66) on the same C-ABI kernels, for callers that drive the module by themselves.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _ramp_up_mo(step, max_step, base_mo=0.9, gamma=0.9):
    """Momentum of the *new* prototype estimate, decaying from ``base_mo`` (aux_path_memory.py:118-120)."""
    return (1 - step / max_step) ** gamma * base_mo


class AuxPath(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.num_classes = kwargs['num_classes']
        self.feat_stage = list(kwargs['feat_stage'])
        self.feat_ch = list(kwargs['feat_ch'])
        self.hid_ch = kwargs['hid_ch']
        self.aux_drop_prob = kwargs['aux_drop_prob']
        if not 0.0 <= self.aux_drop_prob < 1.0:
            raise ValueError(f'dropout probability has to be in [0, 1), got {self.aux_drop_prob}')
        self.layer_bottleneck = nn.Sequential(
            nn.Dropout2d(self.aux_drop_prob),
            nn.Conv2d(sum(self.feat_ch), self.hid_ch, 3, 1, 1),
            nn.BatchNorm2d(self.hid_ch),
            nn.LeakyReLU(1e-2),
        )
        self.fc_cls = nn.Sequential(
            nn.Dropout2d(self.aux_drop_prob),
            nn.Conv2d(self.hid_ch, self.num_classes, 1, bias=False),
        )
        self.do_memory = kwargs['do_memory']
        self.max_step = kwargs['max_step']
        self.momentum = kwargs['update_momentum']
        self.ensemble_mode = kwargs['ensemble_mode']
        if self.ensemble_mode not in ('cosine_similarity', 'mean'):
            raise ValueError(f'unknown ensemble_mode {self.ensemble_mode!r}')
        self.memory_bank = nn.Parameter(torch.zeros((self.num_classes, self.hid_ch, 1, 1), dtype=torch.float32),
                                        requires_grad=False)

    def current_momentum(self, step):
        return _ramp_up_mo(step, self.max_step, self.momentum)

    def forward(self, end_points, scribble, step):
        """Stand-alone call with the reference's signature and return value (models/aux_path_memory.py:46-66): inside
        ``ConsistencyRegulr`` the auxiliary path is part of the engine's fused step and this method is not used; a caller that
        drives the module by itself gets the same arithmetic from the same C-ABI kernels through ONE autograd node (fp32
        convolution kernels, own buffers; gradients flow into the end points and the module's parameters)."""
        feats = [end_points.get(s) for s in self.feat_stage]
        conv, bn, fc = self.layer_bottleneck[1], self.layer_bottleneck[2], self.fc_cls[1]
        la, lm, targets = _AuxFunction.apply(self, scribble, step, *feats, conv.weight, conv.bias, bn.weight, bn.bias, fc.weight)
        out = {'logits_aux_cls': la, 'aux_targets': targets}
        if self.do_memory:
            out.update({'logits_memory': lm,
                        'memory_target': torch.arange(self.num_classes, dtype=torch.long, device=la.device)})
        return out


def _pad4(c):
    return (c + 3) // 4 * 4


class _AuxFunction(torch.autograd.Function):
    """AuxPath.forward as one autograd node over libpacingpseudo_hip.so."""

    @staticmethod
    def forward(ctx, aux, scribble, step, *rest):
        from .._lib import lib, stream_ptr
        nf = len(aux.feat_stage)
        feats, (wc, bc, gamma, beta, wfc) = rest[:nf], rest[nf:]
        conv, bn = aux.layer_bottleneck[1], aux.layer_bottleneck[2]
        if not all(torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32 for f in feats):
            raise TypeError('AuxPath.forward: the end points must be float32 tensors on the GPU')
        st = stream_ptr()
        dev = feats[0].device
        f32 = dict(device=dev, dtype=torch.float32)
        B, _, h, w = feats[0].shape
        K, hid = aux.num_classes, aux.hid_ch
        H, W = scribble.shape[-2:]
        x = torch.cat([f.permute(0, 2, 3, 1) for f in feats], 3).contiguous()           # NHWC (layout change only)
        Cin = x.shape[-1]
        if Cin % 4 or hid % 4:
            raise NotImplementedError('AuxPath.forward: channel counts must be multiples of 4')
        training = aux.training
        p = aux.aux_drop_prob
        keep = 1.0 - p
        drop = None
        if training and p > 0:            # nn.Dropout2d: per-(sample, channel) masks, kept for the backward pass
            def mask(n, c):
                return torch.empty((n, c), **f32).bernoulli_(keep).div_(keep)
            drop = {'input': mask(B, Cin), 'features': mask(B, hid), 'bank': mask(K, hid)}
            xin = torch.empty_like(x)
            lib.pp_channel_scale(x.data_ptr(), Cin, xin.data_ptr(), Cin, drop['input'].data_ptr(), Cin, B, h * w, 0, st)
        else:
            xin = x
        wf = torch.empty((hid, 9, Cin), **f32)
        wb = torch.empty((Cin, 9, hid), **f32)
        lib.pp_pack_conv3x3_weights(wc.data_ptr(), hid, Cin, Cin, wf.data_ptr(), wb.data_ptr(), st)
        z = torch.empty((B, h, w, hid), **f32)
        lib.pp_conv3x3_fwd(xin.data_ptr(), Cin, Cin, wf.data_ptr(), bc.data_ptr(), z.data_ptr(), hid, hid, B, h, w, 1, 0, st)
        coef = torch.empty((4, 1, hid), **f32)
        mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
        P = B * h * w
        nws = max(lib.pp_bn_workspace(hid, P, 1) + 12 * hid, lib.pp_conv3x3_bwd_weight_workspace(hid, Cin, B, h, w),
                  lib.pp_conv1x1_bwd_workspace(K, hid, B, h * w), lib.pp_conv1x1_bwd_workspace(K, hid, 1, K)) + 256
        ws = torch.empty(nws, device=dev, dtype=torch.uint8)
        if training:
            lib.pp_bn_train_stats(z.data_ptr(), hid, hid, P, 1, bn.eps, bn.momentum, gamma.data_ptr(), beta.data_ptr(),
                                  bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                                  mean, invstd, scale, shift, ws.data_ptr(), nws, st)
        else:
            lib.pp_bn_eval_coeffs(hid, 1, bn.eps, gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                                  bn.running_var.data_ptr(), mean, invstd, scale, shift, st)
        slope = aux.layer_bottleneck[3].negative_slope
        feat = torch.empty_like(z)
        lib.pp_bn_lrelu_fwd(z.data_ptr(), hid, scale, shift, feat.data_ptr(), hid, hid, P, 1, slope, st)
        ffc = feat
        if drop is not None:
            ffc = torch.empty_like(feat)
            lib.pp_channel_scale(feat.data_ptr(), hid, ffc.data_ptr(), hid, drop['features'].data_ptr(), hid, B, h * w, 0, st)
        lo = torch.empty((B, K, h, w), **f32)
        lib.pp_conv1x1_nhwc_to_nchw_fwd(ffc.data_ptr(), hid, hid, wfc.data_ptr(), None, lo.data_ptr(), K, B, h * w, st)
        # F.interpolate(size = scribble's, bilinear, align_corners=True) through the NHWC kernel (classes padded to a multiple of 4)
        K4 = _pad4(K)
        lo_l = torch.zeros((B, h, w, K4), **f32)
        lo_l[..., :K] = lo.permute(0, 2, 3, 1)
        up = torch.empty((B, H, W, K4), **f32)
        lib.pp_bilinear_fwd(lo_l.data_ptr(), K4, up.data_ptr(), K4, K4, B, h, w, H, W, st)
        logits_aux = up[..., :K].permute(0, 3, 1, 2).contiguous()
        scb = scribble.to(torch.float32).contiguous()
        targets = torch.empty((B, H, W), device=dev, dtype=torch.int64)
        lib.pp_argmax_channels(scb.data_ptr(), B, scb.shape[1], H * W, targets.data_ptr(), st)
        logits_mem = torch.zeros((K, K, 1, 1), **f32)
        bank_fc = None
        if aux.do_memory:
            bank = aux.memory_bank
            mws = torch.empty(lib.pp_memory_update_workspace(K, hid), device=dev, dtype=torch.uint8)
            lib.pp_memory_update(feat.data_ptr(), hid, hid, h, w, scb.data_ptr(), K, H, W, bank.data_ptr(),
                                 float(aux.current_momentum(step)), 1 if aux.ensemble_mode == 'cosine_similarity' else 0,
                                 mws.data_ptr(), mws.numel(), st)
            bank_fc = bank.detach().reshape(K, hid)
            if drop is not None:
                bank_fc = bank_fc * drop['bank']
            bank_fc = bank_fc.contiguous()
            lm_t = torch.empty((1, K, K), **f32)                     # [class k][bank row p]
            lib.pp_conv1x1_nhwc_to_nchw_fwd(bank_fc.data_ptr(), hid, hid, wfc.data_ptr(), None, lm_t.data_ptr(), K, 1, K, st)
            logits_mem = lm_t[0].t().reshape(K, K, 1, 1).contiguous()
        ctx.aux, ctx.training, ctx.slope, ctx.drop, ctx.shape = aux, training, slope, drop, (B, h, w, H, W, Cin, [f.shape[1] for f in feats])
        ctx.save_for_backward(xin, z, ffc, coef, wb, wfc, gamma, bank_fc if bank_fc is not None else coef, ws)
        ctx.has_bank = bank_fc is not None
        ctx.mark_non_differentiable(targets)
        return logits_aux, logits_mem, targets

    @staticmethod
    def backward(ctx, g_la, g_lm, _g_t):
        from .._lib import lib, stream_ptr
        xin, z, ffc, coef, wb, wfc, gamma, bank_fc, ws = ctx.saved_tensors
        aux = ctx.aux
        B, h, w, H, W, Cin, splits = ctx.shape
        K, hid = aux.num_classes, aux.hid_ch
        st = stream_ptr()
        dev = xin.device
        f32 = dict(device=dev, dtype=torch.float32)
        nws = ws.numel()
        mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
        K4 = _pad4(K)
        g_l = torch.zeros((B, H, W, K4), **f32)
        if g_la is not None:
            g_l[..., :K] = g_la.to(torch.float32).permute(0, 2, 3, 1)
        dlo_l = torch.empty((B, h, w, K4), **f32)
        lib.pp_bilinear_bwd(g_l.data_ptr(), K4, dlo_l.data_ptr(), K4, K4, B, h, w, H, W, 0, st)
        dlo = dlo_l[..., :K].permute(0, 3, 1, 2).contiguous()
        dffc = torch.empty((B, h, w, hid), **f32)
        dwfc = torch.zeros_like(wfc)
        lib.pp_conv1x1_nchw_to_nhwc_bwd(dlo.data_ptr(), ffc.data_ptr(), hid, hid, wfc.data_ptr(), dffc.data_ptr(), hid,
                                        dwfc.data_ptr(), None, K, B, h * w, 0, 0, ws.data_ptr(), nws, st)
        if ctx.has_bank and g_lm is not None:
            dl = g_lm.to(torch.float32).reshape(K, K).t().contiguous().reshape(1, K, K)      # [class k][bank row p]
            lib.pp_conv1x1_nchw_to_nhwc_bwd(dl.data_ptr(), bank_fc.data_ptr(), hid, hid, wfc.data_ptr(), None, 0, dwfc.data_ptr(),
                                            None, K, 1, K, 0, 1, ws.data_ptr(), nws, st)
        dfeat = dffc
        if ctx.drop is not None:
            dfeat = torch.empty_like(dffc)
            lib.pp_channel_scale(dffc.data_ptr(), hid, dfeat.data_ptr(), hid, ctx.drop['features'].data_ptr(), hid, B, h * w, 0, st)
        dz = torch.empty_like(z)
        dgamma, dbeta, dbias = (torch.empty(hid, **f32) for _ in range(3))
        lib.pp_bn_lrelu_bwd(dfeat.data_ptr(), hid, z.data_ptr(), hid, scale, shift, mean, invstd, gamma.data_ptr(),
                            1 if ctx.training else 0, dz.data_ptr(), hid, dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), 0,
                            hid, B * h * w, 1, ctx.slope, ws.data_ptr(), nws, st)
        dwc = torch.empty((hid, Cin, 3, 3), **f32)
        lib.pp_conv3x3_bwd_weight(dz.data_ptr(), hid, hid, xin.data_ptr(), Cin, Cin, Cin, B, h, w, 1, dwc.data_ptr(), 0,
                                  ws.data_ptr(), nws, st)
        dx = torch.empty((B, h, w, Cin), **f32)
        lib.pp_conv3x3_bwd_data(dz.data_ptr(), hid, hid, wb.data_ptr(), dx.data_ptr(), Cin, Cin, B, h, w, 1, 0, st)
        if ctx.drop is not None:
            dxd = torch.empty_like(dx)
            lib.pp_channel_scale(dx.data_ptr(), Cin, dxd.data_ptr(), Cin, ctx.drop['input'].data_ptr(), Cin, B, h * w, 0, st)
            dx = dxd
        dfeats, c0 = [], 0
        for c in splits:
            dfeats.append(dx[..., c0:c0 + c].permute(0, 3, 1, 2).contiguous())
            c0 += c
        return (None, None, None, *dfeats, dwc, dbias, dgamma, dbeta, dwfc)
