"""StepEngine: the static execution plan of one PacingPseudo iteration on one MI355X.

The reference runs its step through PyTorch's dynamic autograd graph, one `aten` op at a time
(train_chaos.py:263-315 -> models/consistency_reglur_memory.py:24-102 -> models/unet.py:62-98).  The network is
fixed, so here the whole step is a *static plan*: every activation, gradient and workspace buffer is laid out once
per input shape in HBM (NHWC, the weak and the strong view back to back along the batch axis), the forward and
the hand-derived backward are fixed sequences of C-ABI kernel launches on the current HIP stream, and weight
gradients land directly in one flat slab that the fused Adam kernel (and the RCCL all-reduce) consume.

Layout decisions
  * activations NHWC fp32; `torch.cat((up(x), skip), 1)` (models/unet.py:151) never runs: producers write
    straight into channel slices of pre-allocated concatenation buffers (pixel stride = lower+skip channels),
    and the matching gradient buffers are filled by the consumers' data-gradient kernels with an
    accumulate flag, so multi-consumer tensors need no separate add.
  * the weak and strong passes of the siamese step (consistency_reglur_memory.py:29,48) share every launch:
    batch = [weak | strong]; BatchNorm statistics / running-stat updates are per group and in order, exactly
    as two module calls would produce; weight gradients of both passes fall out of ONE wgrad reduction.
  * cat5 = [stage6 | stage5] is also the auxiliary path's input (aux_path_memory.py:49), so that concat is
    shared as well.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ._lib import PpBnCoefItem, PpLazyIn, PpPackItem, PpWinoPackItem, lib, lib_for, prof_range, stream_ptr

WINO_ENABLED = os.environ.get('PP_WINO', '1') != '0'      # A/B switch for the Winograd path
# split-fp16 ("f16x3") direct convolution for the non-Winograd layers with at least this many output channels
# (scripts/bench_conv.py --f16x3: 1.05-1.9x the fp32 kernels from 64 outputs up, about equal on the 32-output 256x256
# layers in isolation; on the whole step 32 beats 64 by 0.8 ms, r01 sweep)
F16X3_ENABLED = os.environ.get('PP_F16X3', '1') != '0'
F16X3_MIN_COUT = 32
# BatchNorm fused into the convolution epilogues (train: batch statistics emitted by the conv kernel; eval: scale /
# shift / LeakyReLU applied in the epilogue, one-pass backward from y).  PP_FUSE_BN=0 runs the separate kernels (A/B).
FUSE_BN = os.environ.get('PP_FUSE_BN', '1') != '0'
# train-mode BatchNorm + LeakyReLU applied by the CONSUMER of a tensor while it loads (no bn_lrelu_fwd pass, y never stored);
# PP_LAZY_BN=0 restores the separate apply pass (A/B, same results up to the order of one multiply-add)
LAZY_BN = os.environ.get('PP_LAZY_BN', '1') != '0'
# gradient of nn.MaxPool2d folded into the BatchNorm backward of the layer in front of it (pp_bn_lrelu_bwd[_eval]_pool): no
# separate pp_maxpool2_bwd pass over the skip-gradient buffer.  PP_FUSE_POOL_BWD=0: the separate pass (A/B, same results).
FUSE_POOL_BWD = os.environ.get('PP_FUSE_POOL_BWD', '1') != '0'
# ... and the forward max-pooling folded into the BatchNorm apply pass in front of it (pp_bn_lrelu_fwd_pool, train mode)
FUSE_POOL_FWD = os.environ.get('PP_FUSE_POOL_FWD', '1') != '0'
# weight gradients on a SECOND HIP stream: wgrad(L) needs only dz(L) and x(L) and nothing downstream needs it before the
# optimizer, so it runs beside the critical chain dgrad(L) -> BatchNorm backward(L-1) -> ... (two dz buffers in turn, its own
# workspace; the main stream waits for it at bucket boundaries and at the end of the backward pass).  PP_WGRAD_STREAM=0: one stream.
WGRAD_STREAM = os.environ.get('PP_WGRAD_STREAM', '1') != '0'
# CU budget of the persistent direct weight-gradient kernels while they run on that stream (pp_set_wgrad_cus): with the whole chip
# (256) their first blocks occupy every CU and the data-gradient / BatchNorm chain on the main stream queues behind them; 192 leaves
# it room.  Same-box sweep (r05, profiles/r05_experiments/wgrad_cu_budget_sweep.log): 256 -> 32.43 ms, 224 -> 31.88, 192 -> 31.71,
# 160 -> 32.0, 128 -> 32.09, 96 -> 32.7 (one stream: 32.52).  A budget for the Winograd weight-gradient GEMM (PP_WINO_WGRAD_CUS)
# only lost; the stream's HIP priority has no effect (the device offers two levels and the default is the lower one).
# the auxiliary path's forward (bottleneck conv + BN, classifier, its partial CE, the memory-bank update: ~0.4 ms of small,
# latency-bound launches) on the second stream beside the decoder's forward pass, which does not depend on it
AUX_SIDE = os.environ.get('PP_AUX_SIDE', '1') != '0'
# weight gradient of the FIRST convolution (one input channel, no data gradient wanted) folded into the BatchNorm backward that
# forms its dz (pp_bn_lrelu_bwd[_eval]_wgrad_c1): dz is not written and the conv3x3_c4_wgrad launch -- the last kernel of every
# backward pass, alone on the chip -- does not run.  PP_FUSE_WG1=0: the separate launches (A/B).
FUSE_WG1 = os.environ.get('PP_FUSE_WG1', '1') != '0'
# the per-layer weight packs of a step as one launch per family (pp_*_pack_weights_f16x3_batch); False: one launch per layer
PACK_BATCH = True      # (module attribute, no environment switch: tests flip it to compare with the per-layer calls)
COEF_BATCH = True       # eval-mode BatchNorm coefficient rows of the whole backbone in one launch (module attribute: tests compare both)
WGRAD_CUS_SIDE = int(os.environ.get('PP_WGRAD_CUS_SIDE', '192'))
WGRAD_CUS_FULL = 256                                               # the budget without a second stream: the whole chip
# (Round 4 also built on-load BatchNorm for the Winograd input transform, bilinear x2 up-sampling and max-pooling, and moved the
# data-gradient weight packs to the second stream; all four measured slower or neutral on the benchmark step -- DESIGN.md section 3
# "Round 4", profiles/r04_experiments/ -- and were removed in round 5: kernels, entry points, engine branches and tests.)
# On-load BatchNorm in the patch staging of the two-half halo kernel + the halo-tile weight-gradient kernels, for the MID tensor of a
# DoubleConv (conv -> conv, one consumer): ON.  The P phase of that kernel spends most of its time waiting for memory (r04
# phase trace), so the three VALU operations per element ride along and the mid tensor's bn_lrelu_fwd pass disappears
# (four layers of the benchmark network: enc1 / enc2 / dec2 / dec1).  PP_LAZY_HALO=0 restores the separate pass (A/B).
LAZY_HALO = os.environ.get('PP_LAZY_HALO', '1') != '0'
LAZY_HALO_H16 = False      # the same in 16-bit storage plans: measured a wash (r04 A/B: BatchNorm -0.23 ms, halo +0.20, weight gradients +0.10)
WINO_MIN_CIN = int(os.environ.get('PP_WINO_MIN_CIN', '256'))   # tuning knobs (scripts/bench_wino.py)
WINO_MIN_COUT = 64
SLOPE = 1e-2
BN_EPS = 1e-5
BN_MOM = 0.1
CR_VARIANTS = {'ce_loss': 1, 'l1_loss': 2, 'l2_loss': 3, 'kl_loss': 4}

class View:
    """NHWC view: element (n,y,x,c) lives at ptr + es*(((n*H+y)*W+x)*ld + c), es = bytes per element (4: fp32, 2: the fp16
    tensors of a 16-bit-storage plan).  `base`/`n0`/`c0` remember the
    owning torch tensor and the slice, so tests and debuggers can look at the same memory through torch.

    LAZY tensors (round 4; pp_lazy_in in include/pacingpseudo_hip.h): a buffer that can hold a train-mode BatchNorm layer's
    raw convolution output z instead of y = LeakyReLU(BN(z)) owns coefficient rows `coef` ([groups][3][ld]: scale, shift,
    slope; identity rows for channels with final values) and a flag shared by all its views, set by the forward while the
    buffer is lazy.  Consumers with a *_lazy kernel form read y on the fly (`lazy_arg`)."""
    __slots__ = ('ptr', 'ld', 'C', 'N', 'H', 'W', 'base', 'n0', 'c0', 'coef', 'cptr', 'cg', 'flag', 'es')

    def __init__(self, ptr, ld, C, N, H, W, base=None, n0=0, c0=0, coef=None, cptr=0, cg=1, flag=None, es=4):
        self.ptr, self.ld, self.C, self.N, self.H, self.W = ptr, ld, C, N, H, W
        self.base, self.n0, self.c0 = base, n0, c0
        self.coef, self.cptr, self.cg, self.flag = coef, cptr, cg, flag
        self.es = es

    @property
    def storage(self) -> str:
        """'fp32' / 'fp16' / 'bf16': how the elements of this view are stored (the owning tensor's dtype)."""
        if self.es == 4:
            return 'fp32'
        return 'bf16' if (self.base is not None and self.base.dtype == torch.bfloat16) else 'fp16'

    @property
    def dtype(self):
        return {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[self.storage]

    def torch(self) -> torch.Tensor:
        """(N,H,W,C) strided torch view of this memory (RAW contents: z where the buffer is lazy, see values())."""
        return self.base[self.n0:self.n0 + self.N, :, :, self.c0:self.c0 + self.C]

    @property
    def lazy(self) -> bool:
        return self.flag is not None and self.flag[0]

    def lazy_arg(self):
        """ctypes pp_lazy_in of this view, or None when it holds final values."""
        if not self.lazy:
            return None
        return PpLazyIn(self.cptr, self.ld, self.cg)

    def coef_rows(self) -> torch.Tensor:
        """(groups, 3, C) coefficient rows of this view's channels (a torch view of the owning buffer's rows)."""
        g0 = (self.cptr - self.coef.data_ptr()) // 4 // (3 * self.ld)
        return self.coef[g0:g0 + self.cg, :, self.c0:self.c0 + self.C]

    def values(self) -> torch.Tensor:
        """(N,H,W,C) LOGICAL values: y = lrelu(z * scale + shift) where the buffer is lazy (a test / debugging aid; the
        product path never calls it)."""
        t = self.torch()
        if not self.lazy:
            return t
        out = torch.empty((self.N, self.H, self.W, self.C), device=t.device, dtype=t.dtype)
        lz = self.lazy_arg()
        lib_for(self.storage).pp_lazy_materialize(self.ptr, self.ld, ctypes.byref(lz), out.data_ptr(), self.C, self.C, self.N, self.H * self.W, stream_ptr())
        return out


def _pad4(c):
    return (c + 3) // 4 * 4


def _sub(v: View, c0: int, c: int) -> View:
    """Channel slice [c0, c0+c) of a view."""
    return View(v.ptr + v.es * c0, v.ld, c, v.N, v.H, v.W, v.base, v.n0, v.c0 + c0,
                v.coef, v.cptr + 4 * c0 if v.coef is not None else 0, v.cg, v.flag, v.es)


def _batch(v: View, n0: int, n: int) -> View:
    """Sample range [n0, n0+n) of a view (whole statistics groups when the view can be lazy)."""
    cptr, cg = v.cptr, v.cg
    if v.coef is not None and v.cg > 1:
        per = v.N // v.cg
        assert n0 % per == 0 and n % per == 0, 'a batch slice of a lazy buffer must cover whole statistics groups'
        cptr, cg = v.cptr + 4 * (n0 // per) * 3 * v.ld, n // per
    return View(v.ptr + v.es * n0 * v.H * v.W * v.ld, v.ld, v.C, n, v.H, v.W, v.base, v.n0 + n0, v.c0, v.coef, cptr, cg, v.flag, v.es)


class _Layer:
    """Runtime record of one conv3x3 + BN + LeakyReLU layer."""

    def __init__(self, name, conv, bn, dil):
        self.name, self.conv, self.bn, self.dil = name, conv, bn, dil
        self.stride = int(conv.stride[0])     # 2: first convolution of a down-sampling stage under --is_stride_conv (unet.py:113-116)
        self.cout, self.cin = conv.weight.shape[0], conv.weight.shape[1]
        self.cin_pad = _pad4(self.cin)
        # per-plan state (set by _Plan)
        self.x: Optional[View] = None
        self.y: Optional[View] = None
        self.z = None
        self.coef = None
        self.wf = self.wb = None
        self.groups = 1
        self.wino = False           # set per plan: Winograd F(2x2,3x3) path for this layer


class _Plan:
    """All HBM buffers for one (batch per group, H, W, groups) shape."""

    def __init__(self, eng: 'StepEngine', B: int, H: int, W: int, G: int, trainable: bool = True, storage: str = 'fp32'):
        # trainable = False: a forward-only ("light") plan for no-grad calls -- validation / inference at native slice
        # sizes creates one plan per shape, and those need no gradient buffers, kept Winograd inputs or scratch slabs
        self.B, self.H, self.W, self.G = B, H, W, G
        self.trainable = trainable
        # 16-bit storage (BASELINE config 5): every NHWC activation / activation-gradient buffer of this plan is fp16 and the
        # launches go to the _h16 entry points (include/pacingpseudo_hip_h16.h); weights, logits, statistics, parameter
        # gradients and workspaces stay fp32.  The loss gradients are multiplied by a static power-of-two scale so that the
        # small activation gradients stay inside fp16's normal range; the gradient slab is divided by it after the backward.
        # Round 6: storage 'bf16' (what BASELINE.json configs[4] names) -- the same plan with bfloat16 buffers and the _bf16 entry
        # points.  bfloat16 has fp32's exponent range, so its gradients need no loss scale to stay representable; the scale is
        # kept (same default, exact power of two) because the matrix kernels stage 16-bit operands as fp16 numbers.
        self.storage = storage
        self.h16 = storage != 'fp32'              # "16-bit storage" (either kind): what the layout / kernel-family decisions ask
        self.es = 2 if self.h16 else 4
        self.K = lib_for(storage)
        self.loss_scale = eng.loss_scale if self.h16 else 1.0
        adt = {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}[storage]
        self.act_dtype = adt
        self.Bt = B * G
        self.generation = 0
        dev = eng.device
        self._keep: List[torch.Tensor] = []
        f32 = dict(device=dev, dtype=torch.float32)

        self.lazy_coefs: List[torch.Tensor] = []      # coefficient rows of every buffer that can be lazy
        self.lazy_flags: List[list] = []
        self.lazy_mode = None                         # (backbone BN training, aux BN training) of the last forward
        self.packed_key = None                        # StepEngine._weights_key() of the weights this plan's packs were made from

        def act(n, h, w, c, groups=0):
            """A fresh (n, h, w, c) buffer and its view; groups > 0: the buffer may hold a LAZY tensor (coefficient rows)."""
            t = torch.empty((n, h, w, c), device=dev, dtype=adt)
            self._keep.append(t)
            if not groups:
                return t, View(t.data_ptr(), c, c, n, h, w, t, es=self.es)
            coef = torch.empty((groups, 3, c), **f32)
            flag = [False]
            self.lazy_coefs.append(coef)
            self.lazy_flags.append(flag)
            return t, View(t.data_ptr(), c, c, n, h, w, t, 0, 0, coef, coef.data_ptr(), groups, flag, self.es)

        Bt = self.Bt
        net = eng.backbone
        encs, decs = net.enc_blocks(), net.dec_blocks()
        ch = net.ch_ls
        # spatial size per encoder stage
        sizes = []
        h, w = H, W
        for e in encs:
            if e.pooling is not None or getattr(e, 'stride', 1) == 2:
                if h % 2 or w % 2:
                    raise ValueError(f'input {H}x{W} is not divisible by the encoder stride')
                h, w = h // 2, w // 2
            sizes.append((h, w))
        self.sizes = sizes
        # decoder stage k concatenates up(lower) with encoder stage k's output at that stage's size
        self.cat: Dict[int, View] = {}
        self.dcat: Dict[int, View] = {}
        for k in (5, 4, 3, 2, 1):
            d = decs[k]
            hk, wk = sizes[k - 1]
            _, self.cat[k] = act(Bt, hk, wk, d.up_ch + d.skip_ch, G)
            if trainable:
                _, self.dcat[k] = act(Bt, hk, wk, d.up_ch + d.skip_ch)
        self.x0 = act(Bt, H, W, _pad4(net.input_ch))[1]

        max_elems = 0
        self.layers: List[_Layer] = eng.layers
        self.enc_out: Dict[int, View] = {}
        self.enc_in: Dict[int, View] = {}
        self.pooled: Dict[int, View] = {}
        self.dpooled: Dict[int, View] = {}
        self.mid: Dict[str, View] = {}
        self.zbuf: Dict[str, torch.Tensor] = {}
        self.coef: Dict[str, torch.Tensor] = {}
        self.bn_sums: Dict[str, torch.Tensor] = {}
        self.wf: Dict[str, torch.Tensor] = {}
        self.wb: Dict[str, torch.Tensor] = {}

        self.wino: Dict[str, bool] = {}
        self.f16: Dict[str, bool] = {}
        self.wino16_fwd: Dict[str, bool] = {}
        self.wino16_bwd: Dict[str, bool] = {}
        self.wino16_wg: Dict[str, bool] = {}
        self.amax: Dict[str, torch.Tensor] = {}
        self.vkeep: Dict[str, torch.Tensor] = {}
        self.wino_tile: Dict[str, int] = {}
        self.wino_ws = 0
        self.wg_ws_bytes = 0               # workspace of the weight-gradient calls alone (they may run on a second stream)
        # stride-2 convolutions run as stride-1 convolutions at the input resolution: full-resolution z and (zero-stuffed) dz
        self.zfull: Dict[str, torch.Tensor] = {}
        self.dzfull: Dict[str, torch.Tensor] = {}
        self.ct_ws = 0                     # ConvTranspose2d weight-gradient workspace (--is_trans_conv)

        def conv_select(L: _Layer, h, w):
            """Kernel family of one conv layer at (h, w) -- a pure function of the shape: (Winograd?, tile, split-fp16
            Winograd GEMMs?, split-fp16 direct kernels?)."""
            # Winograd F(4x4,3x3) / F(2x2,3x3) for the wide layers: 4x / 2.25x less MFMA work (measured 1.3-3.2x per
            # layer from 128 input channels up, scripts/bench_wino.py); narrow high-resolution layers stay direct
            use = (WINO_ENABLED and L.cin >= WINO_MIN_CIN and L.cout >= WINO_MIN_COUT and L.cin == L.cin_pad
                   and h % (2 * L.dil) == 0 and w % (2 * L.dil) == 0 and L.stride == 1)
            tile = lib.pp_conv3x3_wino_tile(h, w, L.dil) if use else 0
            # split-fp16 GEMMs on pre-split operands (octets along the GEMM K: 8 channels); forward and weight gradient
            # share the kept transformed input, so they take the same path
            ok16 = bool(use and F16X3_ENABLED and tile == 4 and L.cin % 8 == 0 and L.cout % 8 == 0)
            if self.h16 and use and not ok16:      # 16-bit storage has the split-fp16 F(4x4,3x3) Winograd path only
                use, tile = False, 0
            f16 = bool(F16X3_ENABLED and not use and L.cin_pad == L.cin and L.cin % 4 == 0 and L.cout % 4 == 0
                       and L.cout >= F16X3_MIN_COUT)
            return use, tile, ok16, f16

        # ---- which layer outputs stay LAZY in train-mode BN (static per plan): every consumer of the tensor must have a
        # kernel form that applies BatchNorm + LeakyReLU while loading.  Decided before any buffer is laid out, because a
        # lazy layer writes its raw convolution output z straight into the buffer y would have occupied.
        stages = [int(s_.rsplit('stage', 1)[1]) for s_ in eng.aux.feat_stage] if eng.aux is not None else []
        aux_alias = bool(stages == [6, 5] and decs[5].identity_up and len({sizes[s_ - 1] for s_ in stages}) == 1)
        def halo_lazy_ok(Lc: _Layer, h, w):
            """The direct (two-half halo) forward kernel and the halo-tile weight-gradient kernels of layer Lc read a lazy input."""
            use, _, _, f16 = conv_select(Lc, h, w)
            # (fp32 storage by default: with fp16 tensors the saved pass is half as long -- same-box A/B by family: BatchNorm
            # -0.5 ms, halo +0.16, weight gradients +0.07 with fp32 storage; -0.23 / +0.20 / +0.10 with 16-bit storage, i.e. nothing:
            # engine.LAZY_HALO_H16 switches it on there)
            return bool(LAZY_HALO and (not self.h16 or LAZY_HALO_H16) and not use and f16 and Lc.stride == 1 and G <= 2
                        and self.K.pp_conv3x3_lazy_ok(Lc.cin, Lc.cout, self.Bt, h, w, Lc.dil) == 1)
        self.lazy_out: Dict[str, bool] = self._decide_lazy(eng, encs, decs, sizes, conv_select, stages, aux_alias, halo_lazy_ok)

        def conv_bufs(L: _Layer, n, h, w):
            """Packed-weight buffers of one conv layer and the choice direct vs Winograd."""
            nonlocal max_elems
            use, tile, ok16, f16 = conv_select(L, h, w)
            self.wino[L.name] = use
            self.wino16_fwd[L.name] = self.wino16_bwd[L.name] = self.wino16_wg[L.name] = False
            if use:
                self.wino16_fwd[L.name] = self.wino16_bwd[L.name] = self.wino16_wg[L.name] = ok16
                planes = (tile + 2) ** 2                                        # 16 or 36
                self.wino_tile[L.name] = tile
                self.wf[L.name] = torch.empty((planes, L.cout, L.cin), **f32)   # Uf
                self.wb[L.name] = torch.empty((planes, L.cin, L.cout), **f32)   # Ub
                # transformed input of the forward pass, kept for the weight gradient (2.25-4x the activation, a few
                # GB in total at the benchmark shape: cheaper in 288 GB of HBM than a second transform pass)
                if trainable:
                    self.vkeep[L.name] = torch.empty(lib.pp_conv3x3_wino_vkeep_elems(L.cin, n, h, w, L.dil), **f32)
                    self.wino_ws = max(self.wino_ws, lib.pp_conv3x3_wino_workspace(L.cout, L.cin, n, h, w, L.dil),
                                       lib.pp_conv3x3_wino_bwd_weight_workspace(L.cout, L.cin, n, h, w, L.dil))
                    self.wg_ws_bytes = max(self.wg_ws_bytes, lib.pp_conv3x3_wino_bwd_weight_workspace(L.cout, L.cin, n, h, w, L.dil))
                self.wino_ws = max(self.wino_ws, lib.pp_conv3x3_wino_workspace(L.cin, L.cout, n, h, w, L.dil))
            else:
                self.wf[L.name] = torch.empty((L.cout, 9, L.cin_pad), **f32)
                self.wb[L.name] = torch.empty((L.cin, 9, L.cout), **f32) if L.cin_pad == L.cin else None
            # same buffers hold the [hi4 | lo4] fp16 pairs when the layer runs on the split-fp16 kernels
            self.f16[L.name] = f16
            if self.h16 and not use and not f16 and L.cin_pad == L.cin:
                raise NotImplementedError(f'16-bit storage: {L.name} ({L.cin}->{L.cout}) has no split-fp16 kernel '
                                          f'(needs >= {F16X3_MIN_COUT} output channels)')
            if trainable and (self.f16[L.name] or self.wino16_bwd[L.name] or self.wino16_wg[L.name]):
                self.amax[L.name] = torch.zeros(1, **f32)       # max |dz| of the step, written by the BN backward
            max_elems = max(max_elems, n * h * w * max(L.cout, L.cin_pad))

        self.bn_stats_bytes = 0

        def layer_bufs(L: _Layer, n, h, w, groups):
            self.bn_stats_bytes = max(self.bn_stats_bytes, lib.pp_conv3x3_bn_stats_bytes(L.cout, n, h, w, groups))
            # pre-BatchNorm output z: a lazy layer keeps it in its OUTPUT buffer (train mode) or never stores it (eval mode,
            # fused epilogue), so only the other layers -- and every layer of the unfused A/B path -- own a z buffer
            if not (self.lazy_out[L.name] and FUSE_BN):
                self.zbuf[L.name] = act(n, h, w, L.cout)[0]
            self.coef[L.name] = torch.empty((4, groups, L.cout), **f32)
            # per-channel sums of the split (synchronised) BatchNorm calls: [0] forward, [1] backward local, [2] backward global
            self.bn_sums[L.name] = torch.zeros((3, groups, 2, L.cout), device=dev, dtype=torch.float64)
            if L.stride == 2:                  # (h, w) = output size; the convolution itself runs at (2 h, 2 w)
                self.zfull[L.name] = act(n, 2 * h, 2 * w, L.cout)[0]
                if trainable:
                    self.dzfull[L.name] = act(n, 2 * h, 2 * w, L.cout)[0]
                conv_bufs(L, n, 2 * h, 2 * w)
            else:
                conv_bufs(L, n, h, w)

        cur = self.x0
        for k, e in enumerate(encs, start=1):
            hk, wk = sizes[k - 1]
            if e.pooling is not None:
                self.pooled[k] = act(Bt, hk, wk, cur.C)[1]
                if trainable:
                    self.dpooled[k] = act(Bt, hk, wk, cur.C)[1]
                cur = self.pooled[k]
            self.enc_in[k] = cur
            L1, L2 = eng.enc_layers[k]
            layer_bufs(L1, Bt, hk, wk, G)
            layer_bufs(L2, Bt, hk, wk, G)
            self.mid[L1.name] = act(Bt, hk, wk, L1.cout, G)[1]
            if k <= 5:      # skip slot of decoder stage k
                d = decs[k]
                out = _sub(self.cat[k], d.up_ch, d.skip_ch)
            else:           # stage 6 feeds decoder stage 5 as its lower input
                out = self._lower_slot(5, decs, sizes, act, L2.cout, hk, wk)
            self.enc_out[k] = out
            cur = out
        self.dec_out: Dict[int, View] = {}
        self.gdec: Dict[int, View] = {}
        self.low_src: Dict[int, View] = {5: self.enc_out[6]}
        for k in (5, 4, 3, 2, 1):
            hk, wk = sizes[k - 1]
            L1, L2 = eng.dec_layers[k]
            layer_bufs(L1, Bt, hk, wk, G)
            layer_bufs(L2, Bt, hk, wk, G)
            self.mid[L1.name] = act(Bt, hk, wk, L1.cout, G)[1]
            if k > 1:
                out = self._lower_slot(k - 1, decs, sizes, act, L2.cout, hk, wk)
                self.low_src[k - 1] = out
            else:
                out = act(Bt, hk, wk, L2.cout, G)[1]
            self.dec_out[k] = out
        # gradient wrt each stage output that is NOT a slice of a dcat buffer
        self.g_low: Dict[int, View] = {}
        for k in (5, 4, 3, 2, 1):
            d = decs[k]
            if not d.identity_up and trainable:
                src = self.low_src[k]
                self.g_low[k] = act(Bt, src.H, src.W, src.C)[1]
            if d.trans and trainable:
                self.ct_ws = max(self.ct_ws, lib.pp_convtranspose_bwd_weight_workspace(d.lower_ch, d.skip_ch, d.scale, Bt,
                                                                                       self.low_src[k].H, self.low_src[k].W))
        if trainable:
            self.g_head = act(Bt, H, W, ch[0])[1]
            self.dlogits = torch.empty((Bt, net.num_classes, H, W), **f32)

        # auxiliary path (one group of B samples at the stage-5/6 resolution)
        self.aux = None
        self.aux_error = None
        hs = {sizes[s - 1] for s in stages}
        if len(hs) > 1:
            # the reference fails in torch.cat here (aux_path_memory.py:49) -- but only when the path is used
            self.aux_error = ('Sizes of tensors must match except in dimension 1: the auxiliary path concatenates '
                              f'stages {eng.aux.feat_stage} of different spatial size')
        if eng.aux is not None and self.aux_error is None:
            ax = eng.aux
            ha, wa = sorted(hs)[0]
            LA = eng.aux_layer
            a = dict(h=ha, w=wa, stages=stages)
            a['alias_cat5'] = aux_alias
            if not a['alias_cat5']:
                a['in'] = act(B, ha, wa, LA.cin_pad)[1]
                if trainable:
                    a['din'] = act(B, ha, wa, LA.cin_pad)[1]
            self.bn_stats_bytes = max(self.bn_stats_bytes, lib.pp_conv3x3_bn_stats_bytes(LA.cout, B, ha, wa, 1))
            self.zbuf[LA.name] = act(B, ha, wa, LA.cout)[0]          # the aux features are always materialised (memory_update reads them)
            self.coef[LA.name] = torch.empty((4, 1, LA.cout), **f32)
            self.bn_sums[LA.name] = torch.zeros((3, 1, 2, LA.cout), device=dev, dtype=torch.float64)
            conv_bufs(LA, B, ha, wa)
            a['feat'] = act(B, ha, wa, LA.cout)[1]
            if trainable:
                a['dfeat'] = act(B, ha, wa, LA.cout)[1]
            if ax.aux_drop_prob > 0:        # Dropout2d (aux_path_memory.py:22,31): masked copies + their gradients
                a['drop_in'] = act(B, ha, wa, LA.cin_pad)[1]
                a['drop_feat'] = act(B, ha, wa, LA.cout)[1]
                a['drop_bank'] = torch.empty((ax.num_classes, ax.hid_ch), **f32)
                if trainable:
                    a['drop_din'] = act(B, ha, wa, LA.cin_pad)[1]
                    a['drop_dfeat'] = act(B, ha, wa, LA.cout)[1]
            a['lo'] = torch.empty((B, ax.num_classes, ha, wa), **f32)
            if trainable:
                a['dlo'] = torch.empty((B, ax.num_classes, ha, wa), **f32)
            self.aux = a

        # per-block BatchNorm partial sums emitted by the fused convolution epilogues (one buffer, stream-ordered reuse)
        self.bn_stats = torch.empty(self.bn_stats_bytes // 8 + 2, device=dev, dtype=torch.float64)
        self.rows_out = ctypes.c_int(0)
        # two scratch slabs for the transient gradients (dz of the current layer / dy of the layer below)
        if trainable:
            self.s1 = torch.empty(max_elems, device=dev, dtype=adt)
            self.s2 = torch.empty(max_elems, device=dev, dtype=adt)
            # second dz buffer + events: layer L + 1 writes its dz while the weight gradient of layer L still reads the other one
            self.s1b = torch.empty(max_elems, device=dev, dtype=adt) if WGRAD_STREAM else None
            self.dz_ready = [torch.cuda.Event(), torch.cuda.Event()]
            # (events are created here, not lazily inside a step; wg_pending: recorded during the backward pass in flight --
            # every backward starts with both slots free, the previous one having joined the second stream before it returned,
            # so no step waits on an event of an earlier step and a backward is self-contained for hipGraph capture)
            self.wg_done = [torch.cuda.Event(), torch.cuda.Event()]
            self.wg_pending = [False, False]
            self.dz_slot = 0
            self.bucket_ev = torch.cuda.Event() if WGRAD_STREAM else None      # main -> second stream, at gradient-bucket boundaries

        # workspaces (the weight-gradient query sizes for the largest CU budget a launch may run under, whatever this thread set)
        self.res = None                  # (workspace, bytes, stats buffer, bytes) of the SECOND stream while the aux forward runs there
        self.bn_stats_side = None
        aux_side_ws = 0
        loss_ws = max(lib.pp_seg_losses_workspace(B, H * W), 1024 * 16)
        if eng.aux is not None:
            loss_ws = max(loss_ws, lib.pp_memory_update_workspace(net.num_classes, eng.aux.hid_ch))
        wg = 0
        bn = 0
        for L in eng.layers + ([eng.aux_layer] if self.aux is not None else []):
            n = Bt if L is not eng.aux_layer else B
            g = G if L is not eng.aux_layer else 1
            hL, wL = self._layer_hw(eng, L)
            if trainable:
                wg = max(wg, lib.pp_conv3x3_bwd_weight_workspace(L.cout, L.cin_pad, n, hL, wL))
                self.wg_ws_bytes = max(self.wg_ws_bytes, lib.pp_conv3x3_bwd_weight_workspace(L.cout, L.cin_pad, n, hL, wL))
            bn = max(bn, lib.pp_bn_workspace(L.cout, (n // g) * hL * wL, g) + 12 * g * L.cout)
            if trainable and L.cin == 1:
                bn = max(bn, lib.pp_bn_lrelu_bwd_wgrad_c1_workspace(L.cout, (n // g) * hL * wL, g),
                         lib.pp_bn_lrelu_bwd_wgrad_c1_workspace(L.cout, n * hL * wL, 1))
            if L is eng.aux_layer:      # what the auxiliary forward needs when it runs on the second stream (its own workspace / statistics rows)
                aux_side_ws = max(lib.pp_bn_workspace(L.cout, (n // g) * hL * wL, g) + 12 * g * L.cout, loss_ws,
                                  lib.pp_conv1x1_bwd_workspace(net.num_classes, L.cout, n, hL * wL),
                                  lib.pp_conv3x3_wino_workspace(L.cin, L.cout, n, hL, wL, L.dil) if self.wino[L.name] else 0)
        head = lib.pp_conv1x1_bwd_workspace(net.num_classes, ch[0], Bt, H * W)
        if self.aux is not None:
            head = max(head, lib.pp_conv1x1_bwd_workspace(net.num_classes, eng.aux_layer.cout, B,
                                                          self.aux['h'] * self.aux['w']))
        self.ws_bytes = max(wg, bn, head, loss_ws, self.wino_ws, self.ct_ws) + 256
        self.ws = torch.empty(self.ws_bytes, device=dev, dtype=torch.uint8)
        self.ws_wg = None
        if trainable and WGRAD_STREAM:
            self.wg_ws_bytes = max(self.wg_ws_bytes, aux_side_ws) + 256
            self.ws_wg = torch.empty(self.wg_ws_bytes, device=dev, dtype=torch.uint8)
            if self.aux is not None:
                LA_ = eng.aux_layer
                self.bn_stats_side_bytes = lib.pp_conv3x3_bn_stats_bytes(LA_.cout, B, self.aux['h'], self.aux['w'], 1)
                self.bn_stats_side = torch.empty(self.bn_stats_side_bytes // 8 + 2, device=dev, dtype=torch.float64)
                self.aux_fork, self.aux_join = torch.cuda.Event(), torch.cuda.Event()
        # loss denominators / numerators, packed so that data-parallel runs all-reduce them ONCE per step:
        # [0:6] segmentation losses (pp_seg_losses_fwd), [6:8] auxiliary partial CE (pp_aux_pce_fwd)
        self.all_sums = torch.zeros(8, device=dev, dtype=torch.float64)
        self.sums = self.all_sums[:6]
        if self.aux is not None:
            self.aux['sums'] = self.all_sums[6:8]
        self.target = torch.empty((B, H, W), device=dev, dtype=torch.int64)

    def ws_args(self):
        """(pointer, bytes) of the workspace of the stream that is being launched on (the second stream's while `res` is set)."""
        return (self.res[0].data_ptr(), self.res[1]) if self.res is not None else (self.ws.data_ptr(), self.ws_bytes)

    def stats_args(self):
        """(pointer, bytes) of the buffer the fused convolution epilogues leave their per-block BatchNorm sums in."""
        return (self.res[2].data_ptr(), self.res[3]) if self.res is not None else (self.bn_stats.data_ptr(), self.bn_stats_bytes)

    def begin_forward(self, mode):
        """Start of a forward through this plan: no buffer is lazy yet (the lazy layers of this forward set their flags as
        they run); when the BatchNorm mode differs from the previous forward's, the coefficient rows go back to the identity
        (1, 0, 1), because slices a lazy layer wrote last time may now receive final values."""
        self.generation += 1             # any forward through this plan overwrites its activation buffers
        for f in self.lazy_flags:
            f[0] = False
        if mode != self.lazy_mode:
            for c in self.lazy_coefs:
                c[:, 0].fill_(1.0)
                c[:, 1].zero_()
                c[:, 2].fill_(1.0)
            self.lazy_mode = mode

    def _lower_slot(self, k, decs, sizes, act, c, h, w) -> View:
        """Where the tensor feeding decoder stage k as `lower` is written: straight into cat_k when the
        up-sampling factor is 1 (an exact identity), else into its own buffer (then resized into cat_k)."""
        if decs[k].identity_up:
            return _sub(self.cat[k], 0, c)
        return act(self.Bt, h, w, c, self.G)[1]

    @staticmethod
    def _decide_lazy(eng, encs, decs, sizes, conv_select, stages, aux_alias, halo_lazy_ok=None) -> Dict[str, bool]:
        """{layer name: its output is a lazy tensor in train-mode BN}.  The consumers of a layer output are: the next
        convolution (directly, or through a concatenation buffer it is a slice of), the max-pooling / up-sampling in front
        of the next stage, the 1x1 head, the auxiliary path.  Each must be able to apply BatchNorm + LeakyReLU on load."""
        def hw(L, k):
            h, w = sizes[k - 1]
            return (2 * h, 2 * w) if L.stride == 2 else (h, w)

        def mid_ok(Lc, k):
            """The second convolution of a DoubleConv as the ONLY consumer of the first one's output: the two-half halo kernel +
            halo-tile weight gradient read it lazily (halo_lazy_ok)."""
            return halo_lazy_ok is not None and halo_lazy_ok(Lc, *hw(Lc, k))

        out = {L.name: False for L in eng.layers}
        if eng.aux_layer is not None:
            out[eng.aux_layer.name] = False
        if not (LAZY_BN and FUSE_BN):       # the lazy forms hang off the fused conv + BN entry points
            return out
        # stage outputs (skip connections, pooled / up-sampled / concatenated tensors, the auxiliary input) are always
        # materialised: their consumers (max-pooling, bilinear x2, Winograd input transform) have no on-load form (measured
        # slower, round 4).  Lazy are: the mid tensor of a DoubleConv whose second convolution runs on the halo kernels, and the
        # last decoder output (read by the 1x1 head only: pp_conv1x1_*_lazy).
        for k in range(1, 7):
            L1, L2 = eng.enc_layers[k]
            out[L1.name] = mid_ok(L2, k) and L1.stride == 1
        for k in (5, 4, 3, 2, 1):
            L1, L2 = eng.dec_layers[k]
            out[L1.name] = mid_ok(L2, k)
        out[eng.dec_layers[1][1].name] = True
        return out

    def _layer_hw(self, eng, L):
        if eng.aux is not None and L is eng.aux_layer:
            return self.aux['h'], self.aux['w']
        for k in range(1, 7):
            if L in eng.enc_layers[k]:
                h, w = self.sizes[k - 1]
                return (2 * h, 2 * w) if L.stride == 2 else (h, w)      # the size the convolution kernels run at
        for k in (5, 4, 3, 2, 1):
            if L in eng.dec_layers[k]:
                return self.sizes[k - 1]
        raise KeyError(L.name)


class StepEngine:
    def __init__(self, backbone, aux_path, args):
        self.backbone, self.aux, self.args = backbone, aux_path, args
        self.enc_layers: Dict[int, tuple] = {}
        self.dec_layers: Dict[int, tuple] = {}
        self.layers: List[_Layer] = []
        for k, e in enumerate(backbone.enc_blocks(), start=1):
            cb = e.conv_block
            p = f'enc_block{k}.conv_block.conv_layer'
            pair = (_Layer(p + '1', cb.conv_layer1.conv, cb.conv_layer1.norm_op, e.dilation),
                    _Layer(p + '2', cb.conv_layer2.conv, cb.conv_layer2.norm_op, e.dilation))
            self.enc_layers[k] = pair
            self.layers += list(pair)
        for k, d in backbone.dec_blocks().items():
            cb = d.conv_block
            p = f'dec_block{k}.conv_block.conv_layer'
            pair = (_Layer(p + '1', cb.conv_layer1.conv, cb.conv_layer1.norm_op, 1),
                    _Layer(p + '2', cb.conv_layer2.conv, cb.conv_layer2.norm_op, 1))
            self.dec_layers[k] = pair
            self.layers += list(pair)
        self.aux_layer = None
        if aux_path is not None:
            for s in aux_path.feat_stage:
                if not s.startswith('encoder/stage'):
                    raise NotImplementedError('auxiliary features must be encoder stages')
            self.aux_layer = _Layer('aux.bottleneck', aux_path.layer_bottleneck[1], aux_path.layer_bottleneck[2], 1)
        for L in self.layers[1:] + ([self.aux_layer] if self.aux_layer is not None else []):
            if L.cin % 4 or L.cout % 4:
                raise NotImplementedError(f'{L.name}: channel counts must be multiples of 4 (got {L.cin}->{L.cout})')
        if self.layers[0].cout % 4:
            raise NotImplementedError('init_ch must be a multiple of 4')
        self.plans: 'OrderedDict[tuple, _Plan]' = OrderedDict()      # least recently used first
        self.plans_built = 0
        self.world = 1
        self.rank = 0
        self.comm = None                 # set by pacingpseudo_amd.parallel.attach()
        self.sync_bn = False             # data-parallel only: BatchNorm batch statistics over the GLOBAL batch (epoch 0)
        self.bucket_hook = None          # callable(tag) fired as gradient buckets complete during backward
        self.last = None                 # state saved by forward for backward
        self._rec = None                 # per-layer (x, y, groups) views recorded by the forward in flight
        self.last_drop_masks = None      # Dropout2d masks of the most recent auxiliary forward (tests read them)
        self._bwd_rec = None
        self.last_plan = None            # plan of the most recent forward (tests look at its buffers)
        self._wg_stream = None           # second HIP stream of the weight gradients (created on first use)
        self._coefs_ready = frozenset()  # layers whose eval-mode coefficient rows the running forward has written in one batch
        self._bwd_plan = None            # plan of the backward pass in flight
        # 16-bit storage of activations / activation gradients for TRAINING plans (`--storage fp16`, BASELINE config 5; PP_ACT_H16=1
        # forces it for A/B runs).  Forward-only plans (validation, inference at native slice sizes) stay fp32.
        self.storage = getattr(args, 'storage', 'fp32') or 'fp32'
        if os.environ.get('PP_ACT_H16', '0') == '1' and self.storage == 'fp32':
            self.storage = 'fp16'
        if self.storage not in ('fp32', 'fp16', 'bf16'):
            raise ValueError(f"--storage must be fp32, fp16 or bf16 (got {self.storage!r})")
        self.h16 = self.storage != 'fp32'
        self.loss_scale = float(os.environ.get('PP_LOSS_SCALE', '1024'))      # static, a power of two (exact to remove)
        if self.h16:
            if any(L.stride != 1 for L in self.layers) or any(d.trans for d in backbone.dec_blocks().values()):
                raise NotImplementedError('16-bit storage: the strided / transposed-convolution U-Net variant has no fp16 kernels')
            if not (F16X3_ENABLED and FUSE_BN):
                raise NotImplementedError('16-bit storage needs the split-fp16 matrix kernels and the fused BatchNorm epilogues')

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return self.backbone.final_conv.weight.device

    MAX_TRAIN_PLANS = 2       # a training run has one shape (plus, at most, the ragged last batch)
    MAX_INFER_PLANS = 8       # validation / inference at native slice sizes: one light plan per (batch, H, W)

    def plan_for(self, B, H, W, G, trainable: bool = True) -> _Plan:
        """Buffers for one shape.  Training plans and forward-only plans are cached separately, least recently used out
        first, and the plan of a forward that still awaits its backward is never evicted: a validation epoch over many
        slice sizes cannot push out (and so force the re-allocation of) the multi-GB training plan."""
        storage = self.storage if trainable else 'fp32'
        key = (B, H, W, G, self.device.index, bool(trainable), storage)
        p = self.plans.get(key)
        if p is not None:
            self.plans.move_to_end(key)
            return p
        limit = self.MAX_TRAIN_PLANS if trainable else self.MAX_INFER_PLANS
        same = [k for k in self.plans if k[5] == bool(trainable)]
        live = self.last['plan'] if self.last is not None else None
        for k in same:
            if len(same) < limit:
                break
            if self.plans[k] is not live:
                del self.plans[k]
                same = [q for q in same if q != k]
        p = _Plan(self, B, H, W, G, trainable, storage)
        self.plans_built += 1
        self.plans[key] = p
        return p

    def _check_input(self, t, name):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise TypeError(f'{name} must be a float32 tensor on the GPU (got {t.dtype} on {t.device})')
        return t.contiguous()

    # ------------------------------------------------------------------ layer primitives
    def _weights_key(self):
        """Changes whenever a convolution weight may have changed: the optimizer's slab version (the fused optimizers write
        through raw pointers and bump it), torch's in-place version counters (load_state_dict, user code) and the addresses."""
        ws = [L.conv.weight for L in self.layers] + ([self.aux_layer.conv.weight] if self.aux_layer is not None else [])
        slabs = {id(f): f for f in (getattr(w, '_pp_flat', None) for w in ws) if f is not None}      # FlatSlab back-references
        return (tuple((f.version, f.params._version) for f in slabs.values()), tuple(w._version for w in ws),
                tuple(w.data_ptr() for w in ws))

    def invalidate_packed(self) -> None:
        """Forget which weights the forward-only plans packed.  `_weights_key` sees optimizer steps (slab version), torch's
        in-place version counters and moved storage; writes that bypass all three -- ``p.data.copy_()`` / EMA updates through
        ``.data``, collectives on a view of the slab -- are the CALLER's to announce with this call.  Called by
        pacingpseudo_amd.parallel.attach and by ``load_state_dict`` of ConsistencyRegulr / UNet (a post-hook; so
        inference.load_backbone is covered too).  tests/test_gpu_round5.py: a ``.data`` write + this call re-packs."""
        for p in self.plans.values():
            p.packed_key = None

    def _pack_weights(self, plan: _Plan, st, need_grad: bool = True):
        """Kernel-side weight layouts of every layer (split-fp16 operands, Winograd-domain U).  Forward operands on the main
        stream; the data-gradient operands (Ub / wb: first used much later, in the backward pass) on the second stream when
        there is one.  A forward without gradients whose weights are unchanged since the plan last packed them packs nothing
        (validation / inference: 23 launches per forward, 3 with the batched packs)."""
        key = None
        if not need_grad:
            key = self._weights_key()
            if plan.packed_key == key:
                return
        layers = self.layers + ([self.aux_layer] if (self.aux is not None and plan.aux is not None) else [])
        if PACK_BATCH:
            # one launch per family for all layers (the per-layer launches are 5 - 50 us each behind one another at the start of every
            # step); the item tables are rebuilt when a weight or a packed buffer moved
            bkey = tuple(L.conv.weight.data_ptr() for L in layers)
            if getattr(plan, 'pack_batch_key', None) != bkey:
                direct, wino, single = [], [], []
                for L in layers:
                    wb = plan.wb[L.name]
                    w, wf = L.conv.weight.data_ptr(), plan.wf[L.name].data_ptr()
                    wbp = wb.data_ptr() if wb is not None else None
                    if plan.wino[L.name]:
                        if plan.wino16_fwd[L.name] and plan.wino16_bwd[L.name] and plan.wino_tile[L.name] == 4:
                            wino.append(PpWinoPackItem(w, L.cout, L.cin, wf, wbp))
                        else:
                            single.append(L)
                    elif plan.f16[L.name]:
                        direct.append(PpPackItem(w, L.cout, L.cin, L.cin_pad, wf, wbp))
                    else:
                        single.append(L)
                plan.pack_batch = ((PpPackItem * len(direct))(*direct) if direct else None, len(direct),
                                   (PpWinoPackItem * len(wino))(*wino) if wino else None, len(wino), single)
                plan.pack_batch_key = bkey
            d_arr, d_n, w_arr, w_n, layers = plan.pack_batch
            if d_n:
                lib.pp_pack_conv3x3_weights_f16x3_batch(d_arr, d_n, st)
            if w_n:
                lib.pp_wino_pack_weights_f16x3_batch(w_arr, w_n, st)
        for L in layers:
            wb = plan.wb[L.name]
            w, wf = L.conv.weight.data_ptr(), plan.wf[L.name].data_ptr()
            wbp = wb.data_ptr() if wb is not None else None
            if plan.wino[L.name]:
                tile = plan.wino_tile[L.name]
                f16f, f16b = plan.wino16_fwd[L.name], plan.wino16_bwd[L.name]
                fn_f = plan.K.pp_wino_pack_weights_f16x3 if f16f else plan.K.pp_wino_pack_weights
                fn_b = plan.K.pp_wino_pack_weights_f16x3 if f16b else plan.K.pp_wino_pack_weights
                if fn_f is fn_b:
                    fn_f(w, L.cout, L.cin, tile, wf, wbp, st)
                else:
                    fn_f(w, L.cout, L.cin, tile, wf, None, st)
                    fn_b(w, L.cout, L.cin, tile, None, wbp, st)
            else:
                fn = plan.K.pp_pack_conv3x3_weights_f16x3 if plan.f16[L.name] else plan.K.pp_pack_conv3x3_weights
                fn(w, L.cout, L.cin, L.cin_pad, wf, wbp, st)
        plan.packed_key = key

    def _conv_bn_fused(self, plan, L: _Layer, x: View, out_ptr, ld_out, groups, mode, scale, shift, st):
        """Forward convolution with the BatchNorm side fused into its epilogue (pp_conv3x3[_wino]_fwd_bn); returns the
        number of partial-statistics rows per group (mode 1).  A lazy x is normalised + activated while it is loaded."""
        C = L.cout
        rows = plan.rows_out
        stats, nbytes = plan.stats_args()
        ws_ptr, ws_len = plan.ws_args()
        lz = x.lazy_arg()
        if plan.wino[L.name]:
            vk = plan.vkeep[L.name].data_ptr() if L.name in plan.vkeep else None     # forward-only plans keep no V
            a = (x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), out_ptr, ld_out, C,
                 x.N, x.H, x.W, L.dil, 1 if plan.wino16_fwd[L.name] else 0, vk,
                 ws_ptr, ws_len, mode, scale, shift, SLOPE, groups, stats, nbytes, ctypes.byref(rows))
            assert lz is None, f'{L.name}: the Winograd path has no lazy-input form'
            plan.K.pp_conv3x3_wino_fwd_bn(*a, st)
        else:
            a = (x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), out_ptr, ld_out, C,
                 x.N, x.H, x.W, L.dil, 1 if plan.f16[L.name] else 0, None, mode, scale, shift, SLOPE, groups,
                 stats, nbytes, ctypes.byref(rows))
            if lz is not None:          # (a shape without a lazy form fails inside: plan.lazy_out asked pp_conv3x3_lazy_ok)
                plan.K.pp_conv3x3_fwd_bn_lazy(*a, ctypes.byref(lz), st)
            else:
                plan.K.pp_conv3x3_fwd_bn(*a, st)
        return rows.value

    def _convbn_fwd(self, plan, L: _Layer, x: View, y: View, groups, training, st, pool_out: Optional[View] = None):
        """conv3x3 + BatchNorm + LeakyReLU of one layer.  pool_out: where the 2x2 max-pooled copy of the output goes when the
        caller wants it from the same pass; returns True when it was written (train-mode apply pass), else the caller pools."""
        coef = plan.coef[L.name]
        C = L.cout
        assert x.C == L.cin_pad, (L.name, x.C, L.cin_pad)
        ppg = (x.N // groups) * (x.H // L.stride) * (x.W // L.stride)        # pixels of the OUTPUT per group
        mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
        bn = L.bn
        sync = training and self.comm is not None and self.sync_bn
        # LAZY output (train-mode BN): z goes straight into the buffer of y, the finalize writes the layer's (scale, shift,
        # slope) rows next to it and the separate normalise + activate pass (pp_bn_lrelu_fwd) does not run: the consumers
        # of the tensor evaluate y while they load it (plan.lazy_out guarantees that each of them can)
        lazy = bool(training and plan.lazy_out[L.name])
        zptr, zld = (y.ptr, y.ld) if lazy else ((plan.zbuf[L.name].data_ptr(), C) if L.name in plan.zbuf else (None, C))
        L.x, L.y, L.groups = x, y, groups
        if self._rec is not None:
            # what this step's backward reads (kept with the step, not the layer): input view (+ whether it was lazy),
            # output view, groups, whether the output is lazy (then the output buffer holds z)
            self._rec[L.name] = (x, y, groups, lazy, x.lazy)

        def finalize(sums_ptr, rows, n_per_group):
            args = (sums_ptr, rows, C, n_per_group, groups, BN_EPS, BN_MOM, bn.weight.data_ptr(), bn.bias.data_ptr(),
                    bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                    mean, invstd, scale, shift)
            if lazy:
                plan.K.pp_bn_train_finalize_lazy(*args, y.cptr, y.ld, SLOPE, st)
                y.flag[0] = True
            else:
                plan.K.pp_bn_train_finalize(*args, st)

        if L.stride == 2:
            assert x.lazy_arg() is None and not lazy
            # stride-2 / padding-1 convolution = the stride-1 convolution sampled at the even pixels (pp_spatial.hip)
            zf = plan.zfull[L.name]
            if plan.f16[L.name]:
                plan.K.pp_conv3x3_fwd_f16x3(x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), zf.data_ptr(), C, C,
                                         x.N, x.H, x.W, L.dil, 0, None, st)
            else:
                plan.K.pp_conv3x3_fwd(x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), zf.data_ptr(), C, C,
                                   x.N, x.H, x.W, L.dil, 0, st)
            plan.K.pp_stride2_gather(zf.data_ptr(), C, zptr, C, C, x.N, x.H // 2, x.W // 2, st)
        elif FUSE_BN:
            if training:
                # z + per-block (sum, sum of squares) from the conv epilogue -> finalize -> y = lrelu(z*scale + shift)
                rows = self._conv_bn_fused(plan, L, x, zptr, zld, groups, 1, None, None, st)
                if sync:
                    # reference semantics under sharding: statistics of the WHOLE batch (models/unet.py:189) -- the local
                    # sums are taken again in ONE row per group (the epilogue's per-block rows are not all-reduced)
                    sums = plan.bn_sums[L.name][0]
                    plan.K.pp_bn_stats_sums(zptr, zld, C, ppg, groups, sums.data_ptr(), *plan.ws_args(), st)
                    self.comm.allreduce_sums(sums)
                    finalize(sums.data_ptr(), 1, ppg * self.world)
                else:
                    finalize(plan.stats_args()[0], rows, ppg)
                if not lazy:
                    if pool_out is not None and FUSE_POOL_FWD:
                        plan.K.pp_bn_lrelu_fwd_pool(zptr, zld, scale, shift, y.ptr, y.ld, pool_out.ptr, pool_out.ld, C, y.N, y.H, y.W,
                                                 groups, SLOPE, st)
                        return True
                    plan.K.pp_bn_lrelu_fwd(zptr, zld, scale, shift, y.ptr, y.ld, C, ppg, groups, SLOPE, st)
            else:
                # running statistics are known before the convolution: the epilogue writes y, z never exists.  (The backbone's
                # coefficient rows were written by ONE launch at the start of the forward: _eval_coeffs_batch.)
                if L.name not in self._coefs_ready:
                    plan.K.pp_bn_eval_coeffs(C, groups, BN_EPS, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(), mean, invstd, scale, shift, st)
                self._conv_bn_fused(plan, L, x, y.ptr, y.ld, groups, 2, scale, shift, st)
            return
        elif plan.wino[L.name]:
            assert x.lazy_arg() is None, 'the unfused Winograd call has no lazy-input form (PP_FUSE_BN=0 implies PP_LAZY_BN=0)'
            vk = plan.vkeep[L.name].data_ptr() if L.name in plan.vkeep else None
            fwd = plan.K.pp_conv3x3_wino_fwd_f16x3 if plan.wino16_fwd[L.name] else plan.K.pp_conv3x3_wino_fwd
            fwd(x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), zptr,
                                    zld, C, x.N, x.H, x.W, L.dil, 0, vk, *plan.ws_args(), st)
        elif plan.f16[L.name]:
            assert x.lazy_arg() is None
            plan.K.pp_conv3x3_fwd_f16x3(x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), zptr, zld, C,
                                     x.N, x.H, x.W, L.dil, 0, None, st)
        else:
            assert x.lazy_arg() is None
            plan.K.pp_conv3x3_fwd(x.ptr, x.ld, x.C, plan.wf[L.name].data_ptr(), L.conv.bias.data_ptr(), zptr, zld, C,
                               x.N, x.H, x.W, L.dil, 0, st)
        if sync:
            # reference semantics under sharding: statistics of the WHOLE batch (models/unet.py:189)
            sums = plan.bn_sums[L.name][0]
            plan.K.pp_bn_stats_sums(zptr, zld, C, ppg, groups, sums.data_ptr(), *plan.ws_args(), st)
            self.comm.allreduce_sums(sums)
            finalize(sums.data_ptr(), 1, ppg * self.world)
        elif training:
            assert not lazy
            plan.K.pp_bn_train_stats(zptr, zld, C, ppg, groups, BN_EPS, BN_MOM, bn.weight.data_ptr(),
                                  bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                  bn.num_batches_tracked.data_ptr(), mean, invstd, scale, shift,
                                  *plan.ws_args(), st)
        else:
            plan.K.pp_bn_eval_coeffs(C, groups, BN_EPS, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                  bn.running_mean.data_ptr(), bn.running_var.data_ptr(), mean, invstd, scale, shift, st)
        if not lazy:
            plan.K.pp_bn_lrelu_fwd(zptr, zld, scale, shift, y.ptr, y.ld, C, ppg, groups, SLOPE, st)

    def _convbn_bwd(self, plan, L: _Layer, dy: View, dx: Optional[View], dx_accumulate, training, grads, st, pool: Optional[View] = None):
        """dy: gradient wrt the layer output.  Writes parameter gradients, and dx (+)= data gradient.  pool: gradient of the
        2x2-max-pooled copy of the layer output (half the size of dy), added to each window's winner inside the BatchNorm backward."""
        coef = plan.coef[L.name]
        C = L.cout
        x, y_rec, groups, lazy_out, x_lazy = self._bwd_rec[L.name]
        # pre-BatchNorm output of the forward: in the layer's own z buffer, or -- lazy layer -- in the buffer of its output
        zptr, zld = (y_rec.ptr, y_rec.ld) if lazy_out else ((plan.zbuf[L.name].data_ptr(), C) if L.name in plan.zbuf else (None, C))
        xlz = x.lazy_arg() if (x_lazy and not plan.wino[L.name]) else None      # (Winograd: the kept V was made from y already)
        assert not x_lazy or plan.wino[L.name] or (xlz is not None and plan.f16[L.name]), f'{L.name}: lazy input without a lazy weight gradient'
        ppg = (x.N // groups) * (x.H // L.stride) * (x.W // L.stride)        # pixels of the layer OUTPUT per group
        mean, invstd, scale, shift = (coef[i].data_ptr() for i in range(4))
        gw, gb, gg, gbeta = grads[L.conv.weight], grads[L.conv.bias], grads[L.bn.weight], grads[L.bn.bias]
        if (FUSE_WG1 and dx is None and pool is None and L.cin == 1 and L.stride == 1 and L.dil == 1 and not x_lazy
                and not plan.wino[L.name] and not plan.f16[L.name] and not (training and self.comm is not None and self.sync_bn)):
            # the first layer: dz has one reader, the weight gradient -- formed and consumed in one pass, never written
            if FUSE_BN and not training:
                plan.K.pp_bn_lrelu_bwd_eval_wgrad_c1(dy.ptr, dy.ld, y_rec.ptr, y_rec.ld, scale, L.bn.weight.data_ptr(), L.bn.bias.data_ptr(),
                                                  x.ptr, x.ld, x.H, x.W, gw.data_ptr(), 0, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(),
                                                  0, C, ppg * groups, SLOPE, plan.ws.data_ptr(), plan.ws_bytes, st)
            else:
                plan.K.pp_bn_lrelu_bwd_wgrad_c1(dy.ptr, dy.ld, zptr, zld, scale, shift, mean, invstd, L.bn.weight.data_ptr(),
                                             1 if training else 0, x.ptr, x.ld, x.H, x.W, gw.data_ptr(), 0, gg.data_ptr(),
                                             gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg, groups, SLOPE, plan.ws.data_ptr(),
                                             plan.ws_bytes, st)
            return
        side = self._side_stream(plan)
        slot = 0
        if side is not None:
            slot = plan.dz_slot
            plan.dz_slot ^= 1
            if plan.wg_pending[slot]:                   # the weight gradient that last read this dz buffer (two layers ago)
                torch.cuda.current_stream().wait_event(plan.wg_done[slot])
        dz = (plan.s1b if slot else plan.s1).data_ptr()
        f16 = plan.f16[L.name]
        need_amax = L.name in plan.amax                      # split-fp16 consumers scale dz by a power of two from max |dz|
        if pool is not None:
            am = plan.amax[L.name].data_ptr() if need_amax else None
            y = y_rec
            if FUSE_BN and not training:
                plan.K.pp_bn_lrelu_bwd_eval_pool(dy.ptr, dy.ld, pool.ptr, pool.ld, y.ptr, y.ld, scale, L.bn.weight.data_ptr(),
                                              L.bn.bias.data_ptr(), dz, C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C,
                                              y.N, y.H, y.W, SLOPE, plan.ws.data_ptr(), plan.ws_bytes, am, st)
            else:
                plan.K.pp_bn_lrelu_bwd_pool(dy.ptr, dy.ld, pool.ptr, pool.ld, zptr, zld, scale, shift, mean, invstd,
                                         L.bn.weight.data_ptr(), 1 if training else 0, dz, C, gg.data_ptr(), gbeta.data_ptr(),
                                         gb.data_ptr(), 0, C, y.N, y.H, y.W, groups, SLOPE, plan.ws.data_ptr(), plan.ws_bytes, am, st)
        elif FUSE_BN and not training:
            # eval-mode BN: the forward epilogue wrote y only; one pass over dy and y (pp_bn_lrelu_bwd_eval)
            y = y_rec
            plan.K.pp_bn_lrelu_bwd_eval(dy.ptr, dy.ld, y.ptr, y.ld, scale, L.bn.weight.data_ptr(), L.bn.bias.data_ptr(), dz, C,
                                     gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg * groups, SLOPE,
                                     plan.ws.data_ptr(), plan.ws_bytes, plan.amax[L.name].data_ptr() if need_amax else None, st)
        elif training and self.comm is not None and self.sync_bn:
            loc, glob = plan.bn_sums[L.name][1], plan.bn_sums[L.name][2]
            plan.K.pp_bn_lrelu_bwd_sums(dy.ptr, dy.ld, zptr, zld, scale, shift, mean, invstd, C, ppg, groups, SLOPE,
                                     loc.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, st)
            glob.copy_(loc)
            self.comm.allreduce_sums(glob)
            plan.K.pp_bn_lrelu_bwd_apply(dy.ptr, dy.ld, zptr, zld, scale, shift, mean, invstd, L.bn.weight.data_ptr(), 1,
                                      loc.data_ptr(), glob.data_ptr(), ppg * self.world, dz, C, gg.data_ptr(),
                                      gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg, groups, SLOPE, plan.ws.data_ptr(),
                                      plan.ws_bytes, plan.amax[L.name].data_ptr() if need_amax else None, st)
        elif need_amax:
            plan.K.pp_bn_lrelu_bwd_amax(dy.ptr, dy.ld, zptr, zld, scale, shift, mean, invstd, L.bn.weight.data_ptr(),
                                     1 if training else 0, dz, C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg,
                                     groups, SLOPE, plan.ws.data_ptr(), plan.ws_bytes, plan.amax[L.name].data_ptr(), st)
        else:
            plan.K.pp_bn_lrelu_bwd(dy.ptr, dy.ld, zptr, zld, scale, shift, mean, invstd, L.bn.weight.data_ptr(),
                                1 if training else 0, dz, C, gg.data_ptr(), gbeta.data_ptr(), gb.data_ptr(), 0, C, ppg,
                                groups, SLOPE, plan.ws.data_ptr(), plan.ws_bytes, st)
        if L.stride == 2:
            # gradients of the stride-2 convolution = those of the stride-1 convolution for dz scattered to the even pixels
            dzf = plan.dzfull[L.name]
            plan.K.pp_stride2_scatter(dz, C, dzf.data_ptr(), C, C, x.N, x.H // 2, x.W // 2, st)
            dz = dzf.data_ptr()
        am = plan.amax[L.name].data_ptr() if need_amax else None

        def weight_gradient():
            """On the second stream when there is one (it then gets its own workspace): it waits for everything the main stream
            has enqueued so far."""
            wst, wws, wws_bytes = st, plan.ws.data_ptr(), plan.ws_bytes
            if side is not None:
                plan.dz_ready[slot].record(torch.cuda.current_stream())
                side.wait_event(plan.dz_ready[slot])
                wst, wws, wws_bytes = side.cuda_stream, plan.ws_wg.data_ptr(), plan.wg_ws_bytes
            if plan.wino[L.name]:
                if plan.wino16_wg[L.name]:
                    plan.K.pp_conv3x3_wino_bwd_weight_f16x3(dz, C, C, x.ptr, x.ld, L.cin, x.N, x.H, x.W, L.dil, gw.data_ptr(), 0,
                                                         plan.vkeep[L.name].data_ptr(), wws, wws_bytes, am, wst)
                else:
                    plan.K.pp_conv3x3_wino_bwd_weight(dz, C, C, x.ptr, x.ld, L.cin, x.N, x.H, x.W, L.dil, gw.data_ptr(), 0,
                                                   plan.vkeep[L.name].data_ptr(), wws, wws_bytes, wst)
            elif f16 and xlz is not None:      # x is the raw output of the layer in front: normalised + activated while it is staged
                plan.K.pp_conv3x3_bwd_weight_f16x3_lazy(dz, C, C, x.ptr, x.ld, L.cin_pad, L.cin, x.N, x.H, x.W, L.dil, gw.data_ptr(), 0,
                                                        wws, wws_bytes, plan.amax[L.name].data_ptr(), ctypes.byref(xlz), wst)
            elif f16:     # split-fp16 halo kernel where the shape qualifies, the fp32 kernels otherwise
                plan.K.pp_conv3x3_bwd_weight_f16x3(dz, C, C, x.ptr, x.ld, L.cin_pad, L.cin, x.N, x.H, x.W, L.dil, gw.data_ptr(), 0,
                                                wws, wws_bytes, plan.amax[L.name].data_ptr(), wst)
            else:
                plan.K.pp_conv3x3_bwd_weight(dz, C, C, x.ptr, x.ld, L.cin_pad, L.cin, x.N, x.H, x.W, L.dil, gw.data_ptr(), 0,
                                          wws, wws_bytes, wst)
            if side is not None:
                plan.wg_done[slot].record(side)
                plan.wg_pending[slot] = True

        def data_gradient():
            """The critical chain, always on the main stream."""
            if plan.wino[L.name]:
                if plan.wino16_bwd[L.name]:
                    plan.K.pp_conv3x3_wino_bwd_data_f16x3(dz, C, C, plan.wb[L.name].data_ptr(), dx.ptr, dx.ld, L.cin, x.N, x.H, x.W,
                                                       L.dil, 1 if dx_accumulate else 0, plan.ws.data_ptr(), plan.ws_bytes, am, st)
                else:
                    plan.K.pp_conv3x3_wino_bwd_data(dz, C, C, plan.wb[L.name].data_ptr(), dx.ptr, dx.ld, L.cin, x.N, x.H, x.W,
                                                 L.dil, 1 if dx_accumulate else 0, plan.ws.data_ptr(), plan.ws_bytes, st)
            elif f16:
                plan.K.pp_conv3x3_bwd_data_f16x3(dz, C, C, plan.wb[L.name].data_ptr(), dx.ptr, dx.ld, L.cin, x.N, x.H, x.W, L.dil,
                                              1 if dx_accumulate else 0, plan.amax[L.name].data_ptr(), st)
            else:
                plan.K.pp_conv3x3_bwd_data(dz, C, C, plan.wb[L.name].data_ptr(), dx.ptr, dx.ld, L.cin, x.N, x.H, x.W, L.dil,
                                        1 if dx_accumulate else 0, st)

        # Order on two streams: the weight gradient starts WITH the data gradient.  (Round 5 measured the alternative -- the weight
        # gradient enqueued behind the data gradient, so that it runs beside the HBM-bound BatchNorm backward of the next layer
        # instead of beside another matrix kernel: 31.9 -> 32.9 ms, the BatchNorm family 7.5 -> 9.7 ms.  The direct weight
        # gradients read 2 - 2.5 GB per launch themselves; profiles/r05_experiments/wgrad_behind_dgrad.log.)
        weight_gradient()
        if dx is not None:
            data_gradient()

    def _side_stream(self, plan):
        """The second stream of the weight gradients, or None (switched off, or a plan without the second dz buffer)."""
        if not WGRAD_STREAM or getattr(plan, 's1b', None) is None:
            return None
        if self._wg_stream is None:
            self._wg_stream = torch.cuda.Stream(device=self.device)
        return self._wg_stream

    def _join_side_stream(self, plan):
        """The main stream waits for every weight gradient enqueued so far (before a gradient bucket is handed to the
        all-reduce, and at the end of the backward pass: the optimizer reads the gradients next)."""
        if self._wg_stream is None:
            return
        for slot, ev in enumerate(plan.wg_done):
            if plan.wg_pending[slot]:
                torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------ backbone forward / backward
    def _eval_coeffs_batch(self, plan: _Plan, st):
        """Eval-mode BatchNorm (the reference from epoch 1 on, train_chaos.py:370): scale / shift of every backbone layer depend on
        parameters and running statistics only, so ONE launch at the start of the forward writes all 22 coefficient rows
        (pp_bn_eval_coeffs_batch; bit-identical to the per-layer pp_bn_eval_coeffs it replaces, which sat between the convolutions
        of the critical chain).  The item table is rebuilt when a parameter or a buffer moved."""
        layers = [L for L in self.layers if L.stride == 1]
        key = tuple((L.bn.weight.data_ptr(), L.bn.running_mean.data_ptr(), plan.coef[L.name].data_ptr()) for L in layers)
        if getattr(plan, 'coef_batch_key', None) != key:
            items = []
            for L in layers:
                coef, bn = plan.coef[L.name], L.bn
                items.append(PpBnCoefItem(L.cout, coef.shape[1], bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), *(coef[i].data_ptr() for i in range(4))))
            plan.coef_batch = ((PpBnCoefItem * len(items))(*items), len(items), frozenset(L.name for L in layers))
            plan.coef_batch_key = key
        arr, n, names = plan.coef_batch
        lib.pp_bn_eval_coeffs_batch(arr, n, BN_EPS, st)
        return names

    def _unet_forward(self, plan: _Plan, training, st, logits: torch.Tensor, after_encoder=None):
        """after_encoder: called once every encoder stage has been enqueued (the auxiliary path reads stages 5 / 6: the composite
        step forks it to the second stream there, beside the decoder)."""
        self._coefs_ready = self._eval_coeffs_batch(plan, st) if (FUSE_BN and COEF_BATCH and not training) else frozenset()
        net = self.backbone
        decs = net.dec_blocks()
        G = plan.G
        encs = net.enc_blocks()
        pooled_done = False
        for k, e in enumerate(encs, start=1):
            if e.pooling is not None and not pooled_done:
                src = plan.enc_out[k - 1]
                assert not src.lazy
                plan.K.pp_maxpool2_fwd(src.ptr, src.ld, plan.pooled[k].ptr, plan.pooled[k].ld, src.C, src.N, src.H, src.W, st)
            L1, L2 = self.enc_layers[k]
            self._convbn_fwd(plan, L1, plan.enc_in[k], plan.mid[L1.name], G, training, st)
            # the next stage's max-pooling from the same pass that normalises this stage's output (train mode)
            nxt = plan.pooled.get(k + 1) if (k < 6 and encs[k].pooling is not None) else None
            pooled_done = bool(self._convbn_fwd(plan, L2, plan.mid[L1.name], plan.enc_out[k], G, training, st, pool_out=nxt))
        if after_encoder is not None:
            after_encoder()
        for k in (5, 4, 3, 2, 1):
            d = decs[k]
            cat = plan.cat[k]
            if not d.identity_up:
                src = plan.low_src[k]
                dst = _sub(cat, 0, d.up_ch)
                if d.trans:                   # nn.ConvTranspose2d(lower, skip, k, k, bias=False), unet.py:140,149
                    plan.K.pp_convtranspose_fwd(src.ptr, src.ld, src.C, d.up_samp.weight.data_ptr(), dst.ptr, dst.ld, d.up_ch,
                                             d.scale, src.N, src.H, src.W, st)
                else:
                    assert not src.lazy
                    plan.K.pp_bilinear_fwd(src.ptr, src.ld, dst.ptr, dst.ld, src.C, src.N, src.H, src.W, cat.H, cat.W, st)
            L1, L2 = self.dec_layers[k]
            self._convbn_fwd(plan, L1, cat, plan.mid[L1.name], G, training, st)
            self._convbn_fwd(plan, L2, plan.mid[L1.name], plan.dec_out[k], G, training, st)
        d1 = plan.dec_out[1]
        fc = net.final_conv
        lz = d1.lazy_arg()
        if lz is not None:
            plan.K.pp_conv1x1_nhwc_to_nchw_fwd_lazy(d1.ptr, d1.ld, d1.C, fc.weight.data_ptr(), fc.bias.data_ptr(),
                                                 logits.data_ptr(), net.num_classes, d1.N, d1.H * d1.W, ctypes.byref(lz), st)
        else:
            plan.K.pp_conv1x1_nhwc_to_nchw_fwd(d1.ptr, d1.ld, d1.C, fc.weight.data_ptr(), fc.bias.data_ptr(),
                                            logits.data_ptr(), net.num_classes, d1.N, d1.H * d1.W, st)

    def _unet_backward_decoder(self, plan: _Plan, training, grads, st):
        net = self.backbone
        decs = net.dec_blocks()
        d1 = plan.dec_out[1]
        fc = net.final_conv
        # (the lazy flags still describe the forward being differentiated: any later forward through this plan would have
        # raised in backward_step / unet_backward)
        a = (plan.dlogits.data_ptr(), d1.ptr, d1.ld, d1.C, fc.weight.data_ptr(), plan.g_head.ptr, plan.g_head.ld,
             grads[fc.weight].data_ptr(), grads[fc.bias].data_ptr(), net.num_classes, d1.N, d1.H * d1.W, 0, 0,
             plan.ws.data_ptr(), plan.ws_bytes)
        lz = d1.lazy_arg()
        if lz is not None:
            plan.K.pp_conv1x1_nchw_to_nhwc_bwd_lazy(*a, ctypes.byref(lz), st)
        else:
            plan.K.pp_conv1x1_nchw_to_nhwc_bwd(*a, st)
        g_out = plan.g_head
        for k in (1, 2, 3, 4, 5):
            L1, L2 = self.dec_layers[k]
            m = plan.mid[L1.name]
            dmid = View(plan.s2.data_ptr(), m.C, m.C, m.N, m.H, m.W, plan.s2[:m.N * m.H * m.W * m.C].view(m.N, m.H, m.W, m.C), es=plan.es)
            self._convbn_bwd(plan, L2, g_out, dmid, False, training, grads, st)
            self._convbn_bwd(plan, L1, dmid, plan.dcat[k], False, training, grads, st)
            # gradient wrt the `lower` input of this stage = gradient wrt the previous stage's output
            d = decs[k]
            glow = _sub(plan.dcat[k], 0, d.up_ch)
            if not d.identity_up:
                dst = plan.g_low[k]
                if d.trans:               # ConvTranspose2d: weight gradient from (lower input, d up-sampled), then the data gradient
                    src = plan.low_src[k]
                    plan.K.pp_convtranspose_bwd_weight(glow.ptr, glow.ld, d.up_ch, src.ptr, src.ld, src.C, d.scale, src.N, src.H, src.W,
                                                    grads[d.up_samp.weight].data_ptr(), 0, plan.ws.data_ptr(), plan.ws_bytes, st)
                    plan.K.pp_convtranspose_bwd_data(glow.ptr, glow.ld, d.up_ch, d.up_samp.weight.data_ptr(), dst.ptr, dst.ld, dst.C,
                                                  d.scale, dst.N, dst.H, dst.W, 0, st)
                else:
                    plan.K.pp_bilinear_bwd(glow.ptr, glow.ld, dst.ptr, dst.ld, dst.C, dst.N, dst.H, dst.W, glow.H, glow.W, 0, st)
                glow = dst
            g_out = glow          # for k == 5 this is the gradient wrt encoder stage 6
            if k == 4:
                self._bucket('decoder_upper')
            elif k == 5:
                self._bucket('dec5')
        return g_out

    def _enc_grad_view(self, plan, k, g6):
        if k == 6:
            return g6
        d = self.backbone.dec_blocks()[k]
        return _sub(plan.dcat[k], d.up_ch, d.skip_ch)

    def _unet_backward_encoder(self, plan: _Plan, training, grads, g6: View, st):
        encs = self.backbone.enc_blocks()
        # the pooled-tensor gradient of stage k + 1, when it is folded into this stage's BatchNorm backward instead of being
        # added to the skip-gradient buffer by pp_maxpool2_bwd (not under synchronised BatchNorm: its split calls have no such form)
        fuse_pool = FUSE_POOL_BWD and not (training and self.comm is not None and self.sync_bn)
        pool_grad = None
        for k in (6, 5, 4, 3, 2, 1):
            e = encs[k - 1]
            L1, L2 = self.enc_layers[k]
            g_out = self._enc_grad_view(plan, k, g6)
            m = plan.mid[L1.name]
            dmid = View(plan.s2.data_ptr(), m.C, m.C, m.N, m.H, m.W, plan.s2[:m.N * m.H * m.W * m.C].view(m.N, m.H, m.W, m.C), es=plan.es)
            self._convbn_bwd(plan, L2, g_out, dmid, False, training, grads, st, pool=pool_grad)
            pool_grad = None
            if k == 1:
                self._convbn_bwd(plan, L1, dmid, None, False, training, grads, st)
                self._bucket('enc_rest')
                break
            gprev = self._enc_grad_view(plan, k - 1, g6)
            if e.pooling is not None:
                dp = plan.dpooled[k]
                self._convbn_bwd(plan, L1, dmid, dp, False, training, grads, st)
                src = plan.enc_out[k - 1]
                if fuse_pool and self.enc_layers[k - 1][1].stride == 1:
                    pool_grad = dp               # consumed by the BatchNorm backward of stage k - 1's last layer (next iteration)
                else:
                    plan.K.pp_maxpool2_bwd(src.ptr, src.ld, dp.ptr, dp.ld, gprev.ptr, gprev.ld, src.C, src.N, src.H, src.W, 1, st)
            else:
                self._convbn_bwd(plan, L1, dmid, gprev, True, training, grads, st)
            if k in (6, 5):
                self._bucket(f'enc{k}')

    def _bucket(self, tag):
        """Tell the data-parallel reducer that every gradient of bucket `tag` has been enqueued.

        Round 6: with the weight gradients on the second stream the bucket's all-reduce is issued FROM that stream, which first
        waits for an event of the main stream -- so RCCL's stream orders itself behind both (the bucket's BatchNorm / bias / head
        gradients are main-stream work, its weight gradients second-stream work) and the MAIN stream is not blocked at the bucket
        boundary.  Round 5 made the main stream wait for every pending weight gradient at each of the six boundaries
        (`_join_side_stream`), forfeiting part of the two-stream overlap in data-parallel runs; the
        main stream now waits once, for the collectives themselves, at the end of the backward pass (GradReducer.reduce)."""
        if self.bucket_hook is None:
            return
        plan = self._bwd_plan
        side = self._side_stream(plan) if plan is not None else None
        if side is None or getattr(plan, 'bucket_ev', None) is None:
            if plan is not None:
                self._join_side_stream(plan)
            self.bucket_hook(tag)
            return
        plan.bucket_ev.record(torch.cuda.current_stream())
        side.wait_event(plan.bucket_ev)
        with torch.cuda.stream(side):
            self.bucket_hook(tag)

    # ------------------------------------------------------------------ public: inference of the bare backbone
    @torch.no_grad()
    def infer_end_points(self, x: torch.Tensor, training: bool):
        x = self._check_input(x, 'x')
        B, Cin, H, W = x.shape
        plan = self.plan_for(B, H, W, 1, trainable=False)
        self.last_plan = plan
        plan.begin_forward((bool(training), False))
        self._rec = None
        st = stream_ptr()
        self._pack_weights(plan, st, need_grad=False)
        plan.K.pp_pack_image_nchw_to_nhwc(x.data_ptr(), B, Cin, H, W, plan.x0.ptr, plan.x0.ld, plan.x0.C, st)
        logits = torch.empty((B, self.backbone.num_classes, H, W), device=x.device, dtype=torch.float32)
        self._unet_forward(plan, training, st, logits)
        ep = {'segmentation/logits': logits}
        for k in range(1, 7):
            ep[f'encoder/stage{k}'] = self._as_nchw(plan.enc_out[k])
        for k in (5, 4, 3, 2, 1):
            ep[f'decoder/stage{k}'] = self._as_nchw(plan.dec_out[k])
        return ep

    # ------------------------------------------------------------------ public: the bare backbone, trainable
    def unet_forward_train(self, x: torch.Tensor):
        """UNet.forward with gradients (upper_bound_chaos.py:156): forward in the module's BN mode, state kept for
        unet_backward."""
        x = self._check_input(x, 'x')
        B, Cin, H, W = x.shape
        plan = self.plan_for(B, H, W, 1)
        self.last_plan = plan
        training = self.backbone.training
        plan.begin_forward((bool(training), False))
        self._rec = {}
        st = stream_ptr()
        self._pack_weights(plan, st)
        plan.K.pp_pack_image_nchw_to_nhwc(x.data_ptr(), B, Cin, H, W, plan.x0.ptr, plan.x0.ld, plan.x0.C, st)
        logits = torch.empty((B, self.backbone.num_classes, H, W), device=x.device, dtype=torch.float32)
        self._unet_forward(plan, training, st, logits)
        ep = {'segmentation/logits': logits}
        if self.backbone.elab_end_points:
            for k in range(1, 7):
                ep[f'encoder/stage{k}'] = self._as_nchw(plan.enc_out[k])
            for k in (5, 4, 3, 2, 1):
                ep[f'decoder/stage{k}'] = self._as_nchw(plan.dec_out[k])
        self.last = dict(unet=True, rec=self._rec, gen=plan.generation, plan=plan, bn_training=training)
        return ep

    def unet_backward(self, dlogits: torch.Tensor, grads: Dict[torch.nn.Parameter, torch.Tensor],
                      state: Optional[dict] = None):
        """Backward of unet_forward_train: d loss / d logits (N,K,H,W) -> parameter gradients."""
        S = state if state is not None else self.last
        if S is None or not S.get('unet'):
            raise RuntimeError('unet_backward called without a recorded unet_forward_train')
        if S is self.last:
            self.last = None
        plan: _Plan = S['plan']
        if plan.generation != S['gen']:
            raise RuntimeError('unet_backward: another forward ran through the same buffers after the forward being '
                               'differentiated; call backward before the next forward')
        if tuple(dlogits.shape) != tuple(plan.dlogits.shape):
            raise ValueError(f'gradient of the logits has shape {tuple(dlogits.shape)}, expected {tuple(plan.dlogits.shape)}')
        self._bwd_rec, self._rec = S['rec'], None
        self._bwd_plan = plan
        if getattr(plan, 'wg_pending', None) is not None:
            plan.wg_pending[0] = plan.wg_pending[1] = False
            plan.dz_slot = 0
        st = stream_ptr()
        if plan.loss_scale != 1.0:
            torch.mul(dlogits.to(torch.float32), plan.loss_scale, out=plan.dlogits)
        else:
            plan.dlogits.copy_(dlogits.to(torch.float32))
        with self._wgrad_budget(plan):
            g6 = self._unet_backward_decoder(plan, S['bn_training'], grads, st)
            self._unet_backward_encoder(plan, S['bn_training'], grads, g6, st)
        self._join_side_stream(plan)

    def _wgrad_budget(self, plan):
        """Context: the CU budget of the direct weight-gradient kernels for one backward pass (this thread's launches only --
        the library keeps it per thread), put back on exit so that direct ABI callers on the same thread see what they set."""
        eng = self

        class _Budget:
            def __enter__(self):
                self.prev = lib.pp_get_wgrad_cus()
                lib.pp_set_wgrad_cus(WGRAD_CUS_SIDE if eng._side_stream(plan) is not None else WGRAD_CUS_FULL)

            def __exit__(self, *exc):
                lib.pp_set_wgrad_cus(self.prev)
                return False
        return _Budget()

    def _as_nchw(self, v: View) -> torch.Tensor:
        """Fresh NCHW-shaped (channels-last strided) copy of an engine buffer (a lazy one is normalised + activated on the way)."""
        out = torch.empty((v.N, v.H, v.W, v.C), device=self.device, dtype=v.dtype)
        lz = v.lazy_arg()
        if lz is not None:
            lib_for(v.storage).pp_lazy_materialize(v.ptr, v.ld, ctypes.byref(lz), out.data_ptr(), v.C, v.C, v.N, v.H * v.W, stream_ptr())
        else:
            lib_for(v.storage).pp_copy_slab(v.ptr, v.ld, out.data_ptr(), v.C, v.C, v.N * v.H * v.W, 0, stream_ptr())
        return out.float().permute(0, 3, 1, 2)

    def branch_mask(self, L: _Layer) -> torch.Tensor:
        """(N,C,H,W) bool: the LeakyReLU branch (pre-activation > 0) the kernels took for every output element of layer L in
        the last forward -- read from y, or from z and the layer's coefficients where the output stayed lazy (tests align the
        oracle's non-differentiable choices with the device's through this)."""
        y = L.y
        if y.lazy:
            # through the DEVICE's own expression (pp_lazy_materialize -> pp_lazy_apply4: one fma, as in every consumer and in
            # the BatchNorm backward): a torch restatement rounds twice and disagrees on elements within an ulp of the kink
            tmp = torch.empty((y.N, y.H, y.W, y.C), device=self.device, dtype=y.dtype)
            lz = y.lazy_arg()
            lib_for(y.storage).pp_lazy_materialize(y.ptr, y.ld, ctypes.byref(lz), tmp.data_ptr(), y.C, y.C, y.N, y.H * y.W, stream_ptr())
            m = tmp > 0
        else:
            m = y.torch() > 0
        return m.permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ public: the composite step
    def forward_step(self, batch, mode, step, need_grad: bool):
        """Forward of ConsistencyRegulr (consistency_reglur_memory.py:24-102).  Returns dict of tensors."""
        args = self.args
        image = self._check_input(batch['image'], 'image')
        scribble = self._check_input(batch['scribble'], 'scribble')
        B, Cin, H, W = image.shape
        K = self.backbone.num_classes
        if scribble.shape != (B, K + 1, H, W):
            raise ValueError(f'scribble must be (B,{K + 1},H,W) one-hot, got {tuple(scribble.shape)}')
        train = mode == 'train'
        do_ent = bool(train and args.do_loss_ent)
        do_cr = bool(train and args.do_decoder_consistency)
        do_aux = bool(train and args.do_aux_path)
        do_mem = bool(do_aux and args.do_memory)
        variant = 0
        if do_cr:
            if args.loss_cr_variants not in CR_VARIANTS:
                raise ValueError('The loss is not implemented.')
            variant = CR_VARIANTS[args.loss_cr_variants]
        G = 2 if do_cr else 1
        plan = self.plan_for(B, H, W, G, trainable=need_grad)
        self.last_plan = plan
        if do_aux and plan.aux is None:
            raise RuntimeError(plan.aux_error or 'model was built without an auxiliary path')
        st = stream_ptr()
        bn_training = self.backbone.training
        dev = image.device

        plan.begin_forward((bool(bn_training), bool(self.aux.training) if self.aux is not None else False))
        self._rec = {} if need_grad else None
        with prof_range('pack weights + images'):
            self._pack_weights(plan, st, need_grad=need_grad)
            plan.K.pp_pack_image_nchw_to_nhwc(image.data_ptr(), B, Cin, H, W, plan.x0.ptr, plan.x0.ld, plan.x0.C, st)
            if do_cr:
                strong = self._check_input(batch['image_strong'], 'image_strong')
                x1 = _batch(plan.x0, B, B)
                plan.K.pp_pack_image_nchw_to_nhwc(strong.data_ptr(), B, Cin, H, W, x1.ptr, x1.ld, x1.C, st)
        logits = torch.empty((plan.Bt, K, H, W), device=dev, dtype=torch.float32)
        # class map of the scribbles: the target of the weak and of the auxiliary partial CE (enqueued first: both streams read it)
        plan.K.pp_argmax_channels(scribble.data_ptr(), B, K + 1, H * W, plan.target.data_ptr(), st)
        aux_group = 1 if do_cr else 0      # the aliased end_points dict holds the LAST backbone pass
        A = {}                             # what the auxiliary forward leaves for the rest of this function

        def aux_forward(sa, main_stream=None):
            """The auxiliary head up to its loss sums (aux_path_memory.py:46-66) and the bank update (:68-120), on stream `sa`.  It
            runs before the loss sums are reduced over the ranks, so that ONE all-reduce carries the denominators of all four
            pixel losses (SURVEY.md 8(e) coupling B)."""
            ax, a, LA = self.aux, plan.aux, self.aux_layer
            ain = self._aux_input(plan, aux_group, sa)
            drop = None
            if ax.aux_drop_prob > 0 and ax.training:
                # nn.Dropout2d in front of the bottleneck conv, of the classifier and (through fc_cls) of the memory
                # bank: per-(sample, channel) keep masks scaled by 1/(1-p), drawn from torch's CUDA generator like
                # F.dropout2d does, and kept for the backward pass
                keep = 1.0 - ax.aux_drop_prob

                def mask(n, c):
                    m = torch.empty((n, c), device=dev, dtype=torch.float32).bernoulli_(keep).div_(keep)
                    if main_stream is not None:      # drawn on the second stream, read by the backward pass on the main one
                        m.record_stream(main_stream)
                    return m
                drop = {'input': mask(B, ain.C), 'features': mask(B, LA.cout)}
                if do_mem:
                    drop['bank'] = mask(K, ax.hid_ch)
                din = a['drop_in']
                plan.K.pp_channel_scale(ain.ptr, ain.ld, din.ptr, din.ld, drop['input'].data_ptr(), ain.C, B,
                                     a['h'] * a['w'], 0, sa)
                ain = din
            self.last_drop_masks = drop
            self._convbn_fwd(plan, LA, ain, a['feat'], 1, self.aux.training, sa)
            feat = a['feat']
            ffc = feat                       # what the classifier reads
            if drop is not None:
                ffc = a['drop_feat']
                plan.K.pp_channel_scale(feat.ptr, feat.ld, ffc.ptr, ffc.ld, drop['features'].data_ptr(), feat.C, B,
                                     a['h'] * a['w'], 0, sa)
            wfc = ax.fc_cls[1].weight
            plan.K.pp_conv1x1_nhwc_to_nchw_fwd(ffc.ptr, ffc.ld, ffc.C, wfc.data_ptr(), None, a['lo'].data_ptr(), K, B,
                                            a['h'] * a['w'], sa)
            plan.K.pp_aux_pce_fwd(a['lo'].data_ptr(), B, K, a['h'], a['w'], H, W, plan.target.data_ptr(),
                               args.ignored_index, A['logits_aux'].data_ptr(), a['sums'].data_ptr(), *plan.ws_args(), sa)
            if do_mem and self.rank == 0:
                # only batch sample 0 of the (global) batch updates the bank: aux_path_memory.py:116
                plan.K.pp_memory_update(feat.ptr, feat.ld, feat.C, a['h'], a['w'], scribble.data_ptr(), K, H, W,
                                     ax.memory_bank.data_ptr(), float(ax.current_momentum(step)),
                                     1 if ax.ensemble_mode == 'cosine_similarity' else 0, *plan.ws_args(), sa)
            A.update(drop=drop, feat=feat, wfc=wfc)

        # Round 5: the auxiliary forward on the SECOND stream, forked when the encoder is enqueued (it reads stages 5 / 6) and
        # joined behind the segmentation losses: ~0.4 ms of small, latency-bound launches beside the decoder's forward pass.
        # Own workspace and statistics rows (plan.res); same kernels on the same data: bit-identical to the in-line order.
        side = self._side_stream(plan) if (AUX_SIDE and do_aux and need_grad and plan.bn_stats_side is not None
                                           and not (self.comm is not None and self.sync_bn)) else None
        if do_aux:
            A['logits_aux'] = torch.empty((B, K, H, W), device=dev, dtype=torch.float32)

        def fork_aux():
            main = torch.cuda.current_stream()
            plan.aux_fork.record(main)
            side.wait_event(plan.aux_fork)
            plan.res = (plan.ws_wg, plan.wg_ws_bytes, plan.bn_stats_side, plan.bn_stats_side_bytes)
            try:
                with torch.cuda.stream(side):
                    aux_forward(side.cuda_stream, main)
            finally:
                plan.res = None
            plan.aux_join.record(side)
        with prof_range('forward: weak | strong pass' if do_cr else 'forward: weak pass'):
            self._unet_forward(plan, bn_training, st, logits, after_encoder=fork_aux if side is not None else None)

        valid_mask = batch.get('valid_mask')
        if valid_mask is not None:
            valid_mask = self._check_input(valid_mask, 'valid_mask')
        mask_ptr = valid_mask.data_ptr() if (valid_mask is not None and (do_ent or do_cr)) else None
        zs = logits[B:] if do_cr else None
        plan.K.pp_seg_losses_fwd(logits.data_ptr(), zs.data_ptr() if do_cr else None, plan.target.data_ptr(), mask_ptr,
                              B, K, H * W, args.ignored_index, int(do_ent), variant, plan.sums.data_ptr(),
                              plan.ws.data_ptr(), plan.ws_bytes, st)
        if do_aux:
            if side is not None:
                torch.cuda.current_stream().wait_event(plan.aux_join)
            else:
                aux_forward(st)
            ax, a = self.aux, plan.aux
            drop, feat, wfc, logits_aux = A['drop'], A['feat'], A['wfc'], A['logits_aux']
        if self.comm is not None:
            self.comm.allreduce_sums(plan.all_sums if do_aux else plan.sums)
        out = {}
        loss_pce = torch.empty((), device=dev, dtype=torch.float32)
        loss_ent = torch.empty((), device=dev, dtype=torch.float32) if do_ent else None
        loss_cr = torch.empty((), device=dev, dtype=torch.float32) if do_cr else None
        plan.K.pp_losses_finalize(plan.sums.data_ptr(), 1 if mask_ptr else 0, loss_pce.data_ptr(),
                               loss_ent.data_ptr() if do_ent else None, loss_cr.data_ptr() if do_cr else None, st)
        out['segmentation/logits'] = logits[:B]
        out['loss_pce'] = loss_pce
        if do_ent:
            out['loss_ent'] = loss_ent
        if do_cr:
            out['loss_cr'] = loss_cr
            out['segmentation/logits_strong'] = logits[B:]

        if do_aux:
            loss_aux = torch.empty((), device=dev, dtype=torch.float32)
            plan.K.pp_losses_finalize(a['sums'].data_ptr(), 0, loss_aux.data_ptr(), None, None, st)
            out['logits_aux_cls'] = logits_aux
            out['loss_aux_cls'] = loss_aux
            if do_mem:
                bank = ax.memory_bank          # (updated by rank 0 inside aux_forward)
                if self.comm is not None:
                    self.comm.broadcast_bank(bank)
                loss_mem = torch.empty((), device=dev, dtype=torch.float32)
                bank_fc = bank
                if drop is not None:             # fc_cls(memory_bank) passes through fc_cls's Dropout2d too (aux_path_memory.py:61)
                    bank_fc = a['drop_bank']
                    plan.K.pp_channel_scale(bank.data_ptr(), ax.hid_ch, bank_fc.data_ptr(), ax.hid_ch,
                                         drop['bank'].data_ptr(), ax.hid_ch, K, 1, 0, st)
                plan.K.pp_memory_ce_fwd(bank_fc.data_ptr(), wfc.data_ptr(), K, ax.hid_ch, loss_mem.data_ptr(), st)
                out['loss_memory'] = loss_mem
        if need_grad:
            self.last = dict(rec=self._rec, gen=plan.generation, plan=plan, B=B, H=H, W=W, K=K, logits=logits, mask=valid_mask if mask_ptr else None,
                             do_ent=do_ent, do_cr=do_cr, do_aux=do_aux, do_mem=do_mem, variant=variant,
                             bn_training=bn_training, aux_training=self.aux.training if self.aux is not None else False,
                             aux_group=aux_group, logits_aux=out.get('logits_aux_cls'),
                             drop=self.last_drop_masks if do_aux else None)
        return out

    def _aux_input(self, plan, aux_group, st) -> View:
        a = plan.aux
        B = plan.B
        if a['alias_cat5']:
            return _batch(plan.cat[5], aux_group * B, B)
        c0 = 0
        for s in a['stages']:
            src = _batch(plan.enc_out[s], aux_group * B, B)
            dst = _sub(a['in'], c0, src.C)
            plan.K.pp_copy_slab(src.ptr, src.ld, dst.ptr, dst.ld, src.C, B * src.H * src.W, 0, st)
            c0 += src.C
        return a['in']

    def backward_step(self, g: Dict[str, Optional[torch.Tensor]], grads: Dict[torch.nn.Parameter, torch.Tensor],
                      state: Optional[dict] = None):
        """Backward of the composite step.  g: upstream gradient (0-dim device tensor) per loss name;
        grads: destination tensor per parameter (views of the flat gradient slab); state: the record of the forward
        being differentiated (the autograd node keeps it; default: the most recent forward)."""
        S = state if state is not None else self.last
        if S is None or S.get('unet'):
            raise RuntimeError('backward_step called without a recorded forward_step')
        if S is self.last:
            self.last = None
        plan: _Plan = S['plan']
        if plan.generation != S['gen']:
            raise RuntimeError('backward_step: another forward ran through the same buffers after the forward being '
                               'differentiated (its activations are gone); call backward before the next forward')
        self._bwd_rec = S['rec']
        self._rec = None
        self._bwd_plan = plan
        if getattr(plan, 'wg_pending', None) is not None:
            plan.wg_pending[0] = plan.wg_pending[1] = False
            plan.dz_slot = 0
        st = stream_ptr()
        B = S['B']
        logits = S['logits']

        def gp(name):
            if name in gp_done:
                return gp_done[name]
            t = g.get(name)
            if t is None:
                gp_done[name] = None
                return None
            if not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
                t = t.to(device=logits.device, dtype=torch.float32).reshape(())
            keep.append(t)
            gp_done[name] = t.data_ptr()
            return gp_done[name]
        keep: List[torch.Tensor] = []
        gp_done: Dict[str, Optional[int]] = {}
        mask = S['mask']
        zs_ptr = logits[B:].data_ptr() if S['do_cr'] else None
        dzs_ptr = plan.dlogits[B:].data_ptr() if S['do_cr'] else None
        # Round 5: the head of the auxiliary backward (it needs the loss gradients only) on the second stream, beside the 1x1 head and
        # the first decoder stages of the main chain; the main stream waits for it in front of the auxiliary convolution's backward
        aux_side = None
        if S['do_aux']:
            aux_side = self._side_stream(plan) if (AUX_SIDE and getattr(plan, 'bn_stats_side', None) is not None) else None
            for nm in ('loss_aux_cls', 'loss_memory'):
                gp(nm)                                   # (upstream gradients on the device before the fork)
            if aux_side is not None:
                main = torch.cuda.current_stream()
                plan.aux_fork.record(main)
                aux_side.wait_event(plan.aux_fork)
                plan.res = (plan.ws_wg, plan.wg_ws_bytes, plan.bn_stats_side, plan.bn_stats_side_bytes)
                try:
                    with torch.cuda.stream(aux_side):
                        self._aux_backward_head(plan, S, gp, grads, aux_side.cuda_stream)
                finally:
                    plan.res = None
                plan.aux_join.record(aux_side)
        with self._wgrad_budget(plan):
            self._backward_step_body(plan, S, grads, gp, aux_side, mask, zs_ptr, dzs_ptr, st)
        self._join_side_stream(plan)
        del keep

    def _backward_step_body(self, plan, S, grads, gp, aux_side, mask, zs_ptr, dzs_ptr, st):
        args = self.args
        B, H, W, K = S['B'], S['H'], S['W'], S['K']
        logits = S['logits']
        with prof_range('backward: losses'):
            plan.K.pp_seg_losses_bwd(logits.data_ptr(), zs_ptr, plan.target.data_ptr(), mask.data_ptr() if mask is not None else None,
                                  B, K, H * W, args.ignored_index, int(S['do_ent']), S['variant'],
                                  1 if getattr(args, 'detach_weak_cr', False) else 0, plan.sums.data_ptr(),
                                  gp('loss_pce'), gp('loss_ent'), gp('loss_cr'), plan.loss_scale, plan.dlogits.data_ptr(), dzs_ptr, st)
        with prof_range('backward: decoder'):
            g6 = self._unet_backward_decoder(plan, S['bn_training'], grads, st)
        if S['do_aux']:
            with prof_range('backward: aux path'):
                if aux_side is not None:
                    torch.cuda.current_stream().wait_event(plan.aux_join)
                else:
                    self._aux_backward_head(plan, S, gp, grads, st)
                self._aux_backward_conv(plan, S, grads, st)
                self._bucket('aux')
        with prof_range('backward: encoder'):
            self._unet_backward_encoder(plan, S['bn_training'], grads, g6, st)

    def _aux_backward_head(self, plan, S, gp, grads, st):
        """The part of the auxiliary backward that depends on the loss gradients only: partial CE -> classifier (its weight
        gradient, d features) and the bank CE.  `st` may be the second stream (plan.res then names its workspace)."""
        ax, a = self.aux, plan.aux
        B, H, W, K = S['B'], S['H'], S['W'], S['K']
        wfc = ax.fc_cls[1].weight
        gw = grads[wfc]
        plan.K.pp_aux_pce_bwd(S['logits_aux'].data_ptr(), plan.target.data_ptr(), self.args.ignored_index, gp('loss_aux_cls'), plan.loss_scale,
                           a['sums'].data_ptr(), a['dlo'].data_ptr(), B, K, a['h'], a['w'], H, W, st)
        feat, dfeat = a['feat'], a['dfeat']
        drop = S['drop']
        ffc, dffc = (a['drop_feat'], a['drop_dfeat']) if drop is not None else (feat, dfeat)
        plan.K.pp_conv1x1_nchw_to_nhwc_bwd(a['dlo'].data_ptr(), ffc.ptr, ffc.ld, ffc.C, wfc.data_ptr(), dffc.ptr,
                                        dffc.ld, gw.data_ptr(), None, K, B, a['h'] * a['w'], 0, 0, *plan.ws_args(), st)
        if drop is not None:
            plan.K.pp_channel_scale(dffc.ptr, dffc.ld, dfeat.ptr, dfeat.ld, drop['features'].data_ptr(), feat.C, B,
                                 a['h'] * a['w'], 0, st)
        if S['do_mem']:
            bank_fc = a['drop_bank'] if drop is not None else ax.memory_bank
            plan.K.pp_memory_ce_bwd(bank_fc.data_ptr(), wfc.data_ptr(), K, ax.hid_ch, gp('loss_memory'),
                                 plan.loss_scale / self.world, gw.data_ptr(), 1, st)

    def _aux_backward_conv(self, plan, S, grads, st):
        """Bottleneck convolution + BatchNorm of the auxiliary path backwards, its data gradient added to the stage-5 / 6 gradients
        (behind the decoder's backward pass, which writes them first)."""
        ax, a, LA = self.aux, plan.aux, self.aux_layer
        B = S['B']
        feat, dfeat = a['feat'], a['dfeat']
        drop = S['drop']
        grp = S['aux_group']

        def scatter(din: View):
            """din (B,h,w,Cin) += into the gradient slices of the stages the auxiliary input was gathered from."""
            c0 = 0
            for s in a['stages']:
                dst = _batch(self._enc_grad_view(plan, s, None) if s != 6 else self._stage6_grad(plan), grp * B, B)
                src = _sub(din, c0, dst.C)
                plan.K.pp_copy_slab(src.ptr, src.ld, dst.ptr, dst.ld, dst.C, B * dst.H * dst.W, 1, st)
                c0 += dst.C
        if drop is not None:
            ddin = a['drop_din']
            self._convbn_bwd(plan, LA, dfeat, ddin, False, S['aux_training'], grads, st)
            if a['alias_cat5']:
                dx = _batch(plan.dcat[5], grp * B, B)
                plan.K.pp_channel_scale(ddin.ptr, ddin.ld, dx.ptr, dx.ld, drop['input'].data_ptr(), ddin.C, B,
                                     a['h'] * a['w'], 1, st)
            else:
                plan.K.pp_channel_scale(ddin.ptr, ddin.ld, a['din'].ptr, a['din'].ld, drop['input'].data_ptr(), ddin.C, B,
                                     a['h'] * a['w'], 0, st)
                scatter(a['din'])
        elif a['alias_cat5']:
            dx = _batch(plan.dcat[5], grp * B, B)
            self._convbn_bwd(plan, LA, dfeat, dx, True, S['aux_training'], grads, st)
        else:
            self._convbn_bwd(plan, LA, dfeat, a['din'], False, S['aux_training'], grads, st)
            scatter(a['din'])

    def unscale_grads(self, state, segments, flat=None) -> None:
        """After the backward of `state` (and, data-parallel, after its all-reduce): divide the gradient-slab segments the step
        wrote by the plan's static loss scale (16-bit storage; a power of two, so exact).  A no-op for fp32 plans.  With the
        FlatSlab given, the same pass raises its overflow flag when a gradient is not finite (a static scale can overflow
        fp16 in a bad step) and the fused optimizers then skip the update, as torch.cuda.amp.GradScaler.step would."""
        plan = state['plan']
        if plan.loss_scale == 1.0:
            if flat is not None:
                flat.guard_on = False
            return
        if flat is not None:
            flat.guard[0:1].zero_()
            flat.guard_on = True
        for t in segments:
            if flat is not None:
                lib.pp_scale_guard(t.data_ptr(), t.numel(), 1.0 / plan.loss_scale, flat.guard.data_ptr(), stream_ptr())
            else:
                lib.pp_scale(t.data_ptr(), t.numel(), 1.0 / plan.loss_scale, stream_ptr())

    def set_loss_scale(self, scale: float) -> float:
        """Change the static loss scale of the 16-bit storage mode (train.py halves it when an epoch's updates were mostly
        skipped by the overflow guard): the engine's value for plans still to be built, and every existing 16-bit plan."""
        self.loss_scale = float(scale)
        for p in self.plans.values():
            if p.h16:
                p.loss_scale = self.loss_scale
        return self.loss_scale

    def _stage6_grad(self, plan):
        d = self.backbone.dec_blocks()[5]
        return plan.g_low[5] if not d.identity_up else _sub(plan.dcat[5], 0, d.up_ch)

