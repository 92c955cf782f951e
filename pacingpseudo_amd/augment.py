"""Two-stream augmentation of a whole batch on the GPU.

Host-side mirror of the reference's input pipeline for the training step:
``datasets/chaos/chaos_dataset.py:58-90`` (``CHAOSTwoStream.__getitem__``: base transforms, then a copy through the
strong transforms), ``datasets/chaos/chaos_aug_configs.py:16-86`` (the transform list and its parameters; ACDC / LVSC use
the same list with 224x224 crops and their own class counts) and ``datasets/augmentations.py`` (the transforms).

Split of the work
  * host (this file): the random DECISIONS.  ``draw_sample`` consumes a ``numpy.random.RandomState`` with the same
    calls in the same order as the reference's transforms do on one sample (``uniform() < p`` gates, then the parameter
    draws), and composes Scaling -> RandomRotation -> Mirroring(0) -> Mirroring(1) -> RandomCrop into ONE inverse affine
    map per sample (float64, stored as 12 floats).  Exception to "same calls": the two ``np.random.rand(h, w)`` noise
    fields of ElasticTransform (augmentations.py:259-260) and the ``np.random.normal`` field of GaussianNoise (:365)
    are generated on the device by Philox from ONE host-drawn seed each, so the host stream advances by one ``randint``
    instead of h*w draws there.
  * device (csrc/pp_augment.hip through the C ABI): every per-pixel operation, ~20 small launches per batch.

Where this differs from the reference's pixels (the reference resamples up to three times on the CPU with three
different libraries; cv2 and skimage are not installed in the build image, so those paths cannot be run here --
parity for the interpolating transforms is UNPINNED and stated as such in DESIGN.md):
  * one bicubic resampling (Keys kernel a = -0.75, the kernel of cv2.INTER_CUBIC) of the composed map instead of
    skimage's cubic-spline resize, then scipy's cubic-spline map_coordinates, then cv2's fixed-point warpAffine --
    EXCEPT where ElasticTransform is the only interpolating transform of a sample (no Scaling / RandomRotation drawn): that
    sample is resampled with scipy's own interpolant, the prefiltered cubic B-spline (round 4: pp_aug_spline_prefilter +
    pp_aug_warp_spline), and matches the reference's pixels (tests/test_gpu_augment_ref.py);
    class maps are resampled nearest-neighbour (the reference resizes one-hot planes bilinearly and takes the argmax
    for Scaling, nearest for the other two); no anti-aliasing filter when Scaling shrinks;
  * the elastic displacement field is evaluated on the OUTPUT grid (equal in distribution: the field is stationary and
    isotropic, so moving it across the rotation / mirroring / crop does not change its law);
  * GaussianNoise and the second MeanStdNorm act on the window that survives RandomCrop (the reference: on the whole
    pre-crop image).
Transforms without interpolation -- MeanStdNorm, Mirroring, RandomCrop / padding, valid mask, one-hot, Brightness,
Contrast, GammaAugmentation, GaussianBlur, Mixup's blend -- follow the reference's arithmetic and are tested against a
line-by-line numpy restatement.  The strong view has the reference's four recipes (chaos_aug_configs.py:63-186): colour
only, + GaussianBlur, + Mixup (the partner slice comes from the whole file list, as in the reference, when the loader supplies
candidates -- `train_chaos.py --augmentations TransformsColorMixup` does, round 4 --, else from the same batch), + SimulationLowRes (nearest down, Keys-cubic up, where skimage uses a cubic spline).  Cutout (:23-49) and
Rotation90 (:319-335) are defined in augmentations.py but used by no recipe; they exist here as opt-in transforms
(`p_rot90`, `p_cutout`, both 0 by default): Rotation90 right after RandomRotation (folded into the same affine map, exact),
Cutout as the last transform of the strong view.

Pinned against the reference (round 3): tests/golden/aug_ref.npz holds outputs of the reference's own transforms and of its
CHAOSTwoStream dataset class; tests/test_gpu_augment_ref.py replays the reference's random draws (and, through the
`fields=` argument of `apply`, its ElasticTransform / GaussianNoise fields) through this pipeline and compares pixels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

SKIP = -1.0e30          # "transform not drawn" sentinel of pp_aug_coef
MAP_FLOATS = 12


@dataclass
class AugConfig:
    """Parameters of chaos_aug_configs.py:16-86 (defaults) -- `for_dataset` gives the ACDC / LVSC variants."""
    num_classes: int = 5                       # ignored index = num_classes
    crop_size: Tuple[int, int] = (256, 256)
    scale_range: Tuple[float, float] = (0.7, 1.4)
    p_scaling: float = 0.2
    sigma_range: Tuple[float, float] = (9.0, 13.0)
    alpha_range: Tuple[float, float] = (0.0, 200.0)
    p_elastic: float = 0.2
    degree_range: Tuple[float, float] = (-30.0, 30.0)
    p_rotation: float = 0.2
    p_mirror: float = 0.5
    noise_scale_range: Tuple[float, float] = (0.0, 0.1)
    p_noise: float = 0.15
    strength: float = 1.0                      # TransformsColor(strength), chaos_aug_configs.py:64-86
    p_color: float = 0.8
    do_strong: bool = True
    # strong-view recipe (--augmentations): TransformsColor | TransformsColorBlur | TransformsColorMixup | TransformsColorLow
    recipe: str = 'TransformsColor'
    blur_range: Tuple[float, float] = (1.0, 1.5)        # chaos_aug_configs.py:111
    lam_range: Tuple[float, float] = (0.8, 1.0)          # :136
    lowres_range: Tuple[float, float] = (1.5, 2.0)       # :185
    p_extra: float = 0.8
    # transforms of augmentations.py no recipe uses (opt-in)
    p_rot90: float = 0.0                                 # Rotation90 :319-335, rot_choices (1, 2, 3)
    elastic_spline: bool = True                          # ElasticTransform resamples with scipy's cubic B-spline (:270); False: the Keys kernel of round 3
    p_cutout: float = 0.0                                # Cutout :23-49
    cutout_length: int = 32

    @classmethod
    def for_dataset(cls, name: str, **kw):
        preset = {'chaos': dict(num_classes=5, crop_size=(256, 256)), 'acdc': dict(num_classes=4, crop_size=(224, 224)),
                  'lvsc': dict(num_classes=2, crop_size=(224, 224))}[name]
        preset.update(kw)
        return cls(**preset)


def draw_sample(rng: np.random.RandomState, h: int, w: int, cfg: AugConfig, n_partners: int = 0) -> dict:
    """The random decisions for one h x w slice, in the reference's call order (see the module docstring)."""
    p = dict(h=h, w=w, nh=h, nw=w, scale=None, sigma=0.0, alpha=0.0, field_seed=0, degree=None, flip0=False, flip1=False,
             noise=0.0, noise_seed=0, bright=SKIP, contrast=SKIP, gamma=SKIP, blur=0.0, lam=-1.0, partner=-1, lowres=0.0,
             rot90=0, cutout=None)
    if rng.uniform() < cfg.p_scaling:                                        # Scaling, augmentations.py:200-208
        p['scale'] = rng.uniform(*cfg.scale_range)
        p['nh'], p['nw'] = round(p['scale'] * h), round(p['scale'] * w)
    if rng.uniform() < cfg.p_elastic:                                        # ElasticTransform, :248-260
        p['sigma'] = rng.uniform(*cfg.sigma_range)
        p['alpha'] = rng.uniform(*cfg.alpha_range)
        p['field_seed'] = int(rng.randint(2 ** 31 - 1))
    if rng.uniform() < cfg.p_rotation:                                       # RandomRotation, :299-305
        p['degree'] = rng.uniform(*cfg.degree_range)
    if cfg.p_rot90 > 0 and rng.uniform() < cfg.p_rot90:                      # Rotation90 (opt-in), :326-333
        p['rot90'] = int(rng.randint(3)) + 1                                 # np.random.choice((1, 2, 3))
        if p['rot90'] % 2:
            p['nh'], p['nw'] = p['nw'], p['nh']
    p['flip0'] = bool(rng.uniform() < cfg.p_mirror)                          # Mirroring(axis=0), :343
    p['flip1'] = bool(rng.uniform() < cfg.p_mirror)                          # Mirroring(axis=1)
    if rng.uniform() < cfg.p_noise:                                          # GaussianNoise, :360-365
        p['noise'] = rng.uniform(*cfg.noise_scale_range)
        p['noise_seed'] = int(rng.randint(2 ** 31 - 1))
    rng.uniform()                                                            # RandomCrop's own gate (p = 1), :377
    ch, cw = cfg.crop_size
    w_margin, h_margin = p['nw'] - cw, p['nh'] - ch
    if w_margin > 0:                                                         # :386-397, width first, then height
        p['image_left'], p['canvas_left'] = int(rng.randint(w_margin + 1)), 0
    else:
        p['image_left'], p['canvas_left'] = 0, int(rng.randint(abs(w_margin) + 1))
    if h_margin > 0:
        p['image_top'], p['canvas_top'] = int(rng.randint(h_margin + 1)), 0
    else:
        p['image_top'], p['canvas_top'] = 0, int(rng.randint(abs(h_margin) + 1))
    p['patch_h'], p['patch_w'] = min(p['nh'], ch), min(p['nw'], cw)
    if cfg.do_strong:                                                        # TransformsColor, chaos_aug_configs.py:71-86
        s = cfg.strength
        if rng.uniform() < cfg.p_color:                                      # Brightness, :103-109
            p['bright'] = rng.uniform(-s * 0.8, s * 0.8)
        if rng.uniform() < cfg.p_color:                                      # Contrast, :120-124
            p['contrast'] = rng.uniform(max(0.0, 1 - s * 0.8), 1 + s * 0.8)
        if rng.uniform() < cfg.p_color:                                      # GammaAugmentation, :141-156
            lo, hi = max(0.0, 1 - s * 0.8), 1 + s * 0.8
            if rng.uniform() < 0.5 and lo < 1.0:
                p['gamma'] = rng.uniform(lo, 1.0)
            else:
                p['gamma'] = rng.uniform(max(1.0, lo), hi)
        # the fourth transform of the Blur / Mixup / Low recipes (chaos_aug_configs.py:88-186)
        if cfg.recipe == 'TransformsColorBlur' and rng.uniform() < cfg.p_extra:          # GaussianBlur, augmentations.py:88-94
            p['blur'] = rng.uniform(*cfg.blur_range)
        elif cfg.recipe == 'TransformsColorMixup' and rng.uniform() < cfg.p_extra:       # Mixup, :60-64
            p['lam'] = rng.uniform(*cfg.lam_range)
            p['partner'] = int(rng.randint(n_partners)) if n_partners > 0 else -1       # np.random.choice(file_ls)
        elif cfg.recipe == 'TransformsColorLow' and rng.uniform() < cfg.p_extra:         # SimulationLowRes, :173-177
            p['lowres'] = rng.uniform(*cfg.lowres_range)
        if cfg.p_cutout > 0 and rng.uniform() < cfg.p_cutout:                            # Cutout (opt-in), :30-46
            p['cutout'] = cutout_rect(int(rng.randint(ch)), int(rng.randint(cw)), cfg.cutout_length, ch, cw)
    return p


def cutout_rect(y: int, x: int, length: int, h: int, w: int):
    """(top, left, height, width) of the square Cutout zeroes around the drawn centre (augmentations.py:39-44)."""
    y1, y2 = min(max(y - length // 2, 0), h), min(max(y + length // 2, 0), h)
    x1, x2 = min(max(x - length // 2, 0), w), min(max(x + length // 2, 0), w)
    return (y1, x1, y2 - y1, x2 - x1)


def compose_map(p: dict) -> np.ndarray:
    """12 floats for pp_aug_warp: the inverse map output pixel -> source pixel, in (y, x, 1) coordinates.

    output --RandomCrop^-1--> scaled canvas --Mirroring^-1--> --RandomRotation^-1--> --Scaling^-1--> source slice."""
    h, w, nh, nw = p['h'], p['w'], p['nh'], p['nw']
    crop = np.array([[1, 0, p['image_top'] - p['canvas_top']], [0, 1, p['image_left'] - p['canvas_left']], [0, 0, 1]], np.float64)
    flip = np.eye(3)
    if p['flip0']:
        flip = np.array([[-1, 0, nh - 1], [0, 1, 0], [0, 0, 1]], np.float64) @ flip
    if p['flip1']:
        flip = np.array([[1, 0, 0], [0, -1, nw - 1], [0, 0, 1]], np.float64) @ flip
    rot = np.eye(3)
    if p['degree'] is not None:
        # cv2.getRotationMatrix2D(center=(w/2, h/2), angle) maps source -> destination; warpAffine samples the source
        # at its inverse: x_s = a (x_d - cx) - b (y_d - cy) + cx, y_s = b (x_d - cx) + a (y_d - cy) + cy
        a, b = math.cos(math.radians(p['degree'])), math.sin(math.radians(p['degree']))
        cy, cx = nh / 2.0, nw / 2.0
        rot = np.array([[a, b, cy - a * cy - b * cx], [-b, a, cx + b * cy - a * cx], [0, 0, 1]], np.float64)
    # Rotation90 (np.rot90 by k quarter turns, axes (0, 1)): out[i, j] = in[j, W - 1 - i] per turn, W = the width before
    # that turn; (qh, qw) is the size before all turns (= the size RandomRotation / Scaling left)
    k = p.get('rot90', 0) % 4
    qh, qw = (nw, nh) if k % 2 else (nh, nw)
    r90 = np.eye(3)
    ch_, cw_ = qh, qw
    for _ in range(k):
        r90 = r90 @ np.array([[0, 1, 0], [-1, 0, cw_ - 1], [0, 0, 1]], np.float64)
        ch_, cw_ = cw_, ch_
    if p['degree'] is not None and k:
        cy, cx = qh / 2.0, qw / 2.0
        rot = np.array([[a, b, cy - a * cy - b * cx], [-b, a, cx + b * cy - a * cx], [0, 0, 1]], np.float64)
    # skimage.transform.resize: output pixel centre (i + 0.5) * (h / nh) - 0.5 in source coordinates
    sy, sx = h / qh, w / qw
    scale = np.array([[sy, 0, 0.5 * sy - 0.5], [0, sx, 0.5 * sx - 0.5], [0, 0, 1]], np.float64)
    A = scale @ rot @ r90 @ flip @ crop
    return np.array([A[0, 0], A[0, 1], A[0, 2], A[1, 0], A[1, 1], A[1, 2], p['canvas_top'], p['canvas_left'],
                     p['patch_h'], p['patch_w'], h, w], np.float32)


def pack_params(samples: Sequence[dict]) -> dict:
    """Per-batch arrays (numpy) the device calls consume."""
    B = len(samples)
    out = dict(maps=np.stack([compose_map(p) for p in samples]).astype(np.float32),
               src_rect=np.array([[0, 0, p['h'], p['w']] for p in samples], np.int32),
               out_rect=np.array([[p['canvas_top'], p['canvas_left'], p['patch_h'], p['patch_w']] for p in samples], np.int32),
               # displacement in SOURCE pixels: the field lives in the scaled domain, one source pixel = nh / h of its pixels
               sigma_alpha=np.array([[p['sigma'], p['alpha'] * p['h'] / (p['nw'] if p.get('rot90', 0) % 2 else p['nh'])]
                                     for p in samples], np.float32),
               noise=np.array([p['noise'] for p in samples], np.float32),
               bright=np.array([p['bright'] for p in samples], np.float32),
               contrast=np.array([p['contrast'] for p in samples], np.float32),
               gamma=np.array([p['gamma'] for p in samples], np.float32),
               blur=np.array([[p['blur'], 0.0] for p in samples], np.float32),
               lam=np.array([p['lam'] if p['partner'] >= 0 else -1.0 for p in samples], np.float32),
               partner=np.array([p['partner'] for p in samples], np.int64),
               lowres=np.array([p['lowres'] for p in samples], np.float32),
               cutout=np.array([p['cutout'] if p.get('cutout') is not None else (0, 0, 0, 0) for p in samples], np.int32),
               # ElasticTransform is the sample's only interpolating transform (no Scaling / RandomRotation / Rotation90 drawn):
               # the image is resampled with scipy's cubic B-spline, as the reference does (augmentations.py:270)
               spline=np.array([1 if (p['sigma'] > 0 and p['scale'] is None and p['degree'] is None and not p.get('rot90'))
                                else 0 for p in samples], np.int32))
    # one Philox key per batch: the first drawn seed (samples are distinguished by the counter)
    fs = [p['field_seed'] for p in samples if p['sigma'] > 0]
    ns = [p['noise_seed'] for p in samples if p['noise'] > 0]
    out['field_seed'] = fs[0] if fs else 0
    out['noise_seed'] = ns[0] if ns else 0
    assert out['maps'].shape == (B, MAP_FLOATS)
    return out


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


class DeviceAugmenter:
    """``aug(image, label, scribble[, sizes])`` -> the dict one CHAOSTwoStream batch would hold, all on the GPU.

    image: (B, Hp, Wp) float, label / scribble: (B, Hp, Wp) integer class maps (the reference's ``img`` / ``lab`` / ``scb``
    arrays, chaos_dataset.py:92-105), un-augmented, zero-padded to one plane size when slices differ; ``sizes``: the (h, w)
    of each slice (default: the whole plane)."""

    def __init__(self, cfg: AugConfig, device='cuda', seed: int = 1):
        self.cfg, self.device = cfg, torch.device(device)
        self.rng = np.random.RandomState(seed)
        self.lib = _lib.lib
        self.last_params = None

    # Parameter uploads go through a small ring of PINNED staging slabs owned by the augmenter: a copy from pageable memory is
    # synchronous (it waits for everything the stream still holds, i.e. the previous training step, so the host never ran ahead of
    # the GPU in the training loop), and `Tensor.pin_memory()` per array costs 0.5 ms of host time each, a dozen times per batch
    # (round 5: the driver's host side 8 -> 2 ms per iteration).  A slab is reused after _PIN_SLOTS batches; the event recorded
    # behind its last copy is waited for first (long since complete).
    _PIN_SLOTS, _PIN_BYTES = 8, 1 << 20

    def _pin_begin(self):
        """Start a new batch: take the next staging slab."""
        if self.device.type != 'cuda':
            return
        if getattr(self, '_pin', None) is None:
            self._pin = [torch.empty(self._PIN_BYTES, dtype=torch.uint8).pin_memory() for _ in range(self._PIN_SLOTS)]
            self._pin_np = [t.numpy() for t in self._pin]
            self._pin_ev = [None] * self._PIN_SLOTS
            self._pin_i, self._pin_off = -1, 0
        if self._pin_i >= 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pin_ev[self._pin_i] = ev
        self._pin_i = (self._pin_i + 1) % self._PIN_SLOTS
        self._pin_off = 0
        if self._pin_ev[self._pin_i] is not None:
            self._pin_ev[self._pin_i].synchronize()

    def _up(self, a, dtype):
        np_dt = {torch.float32: np.float32, torch.int32: np.int32, torch.float64: np.float64, torch.int64: np.int64, None: None}[dtype]
        a = np.ascontiguousarray(a, dtype=np_dt)
        if self.device.type != 'cuda' or getattr(self, '_pin', None) is None or a.nbytes == 0:
            return torch.from_numpy(a).to(self.device, non_blocking=True)
        off = (self._pin_off + 15) & ~15
        if off + a.nbytes > self._PIN_BYTES:                 # (does not happen with the recipe's parameter tables: a few KB per batch)
            return torch.from_numpy(a).pin_memory().to(self.device, non_blocking=True)
        self._pin_off = off + a.nbytes
        self._pin_np[self._pin_i][off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        host = self._pin[self._pin_i][off:off + a.nbytes].view(torch.from_numpy(a).dtype).view(a.shape)
        return host.to(self.device, non_blocking=True)

    def draw(self, sizes):
        return [draw_sample(self.rng, int(h), int(w), self.cfg, n_partners=len(sizes)) for h, w in sizes]

    def apply(self, image: torch.Tensor, label: torch.Tensor, scribble: torch.Tensor, samples: Sequence[dict],
              fields: Optional[dict] = None, mix=None) -> dict:
        """`fields` (optional) supplies the two random FIELDS from outside instead of the device's Philox streams:
        'disp' (B, 2, Ho, Wo) displacement in source pixels on the output grid (rows, columns), 'noise' (B, Ho, Wo) the
        additive noise, already scaled.  Used to replay the reference's own draws (tests/test_gpu_augment_ref.py)."""
        cfg, L = self.cfg, self.lib
        B, Hp, Wp = image.shape
        Ho, Wo = cfg.crop_size
        K = cfg.num_classes
        st = torch.cuda.current_stream(self.device).cuda_stream
        self._pin_begin()
        pk = pack_params(samples)
        f32, i32 = torch.float32, torch.int32
        maps, src_rect, out_rect = self._up(pk['maps'], f32), self._up(pk['src_rect'], i32), self._up(pk['out_rect'], i32)
        # non_blocking: the loader hands over PINNED tensors; a blocking copy waits for everything the stream still holds -- the whole
        # previous training step -- and the host never runs ahead of the GPU (round 5: the driver's iteration 33 -> 30 ms)
        img = image.to(self.device, f32, non_blocking=True).contiguous().clone()
        lab = label.to(self.device, i32, non_blocking=True).contiguous()
        scb = scribble.to(self.device, i32, non_blocking=True).contiguous()
        stats = torch.empty(B, 4, device=self.device, dtype=torch.float64)
        stats0 = torch.empty_like(stats)
        coef = torch.empty(B, 4, device=self.device, dtype=f32)

        def norm(x, H, W, rect):                                            # MeanStdNorm
            L.pp_aug_stats(_ptr(x), B, H, W, _ptr(rect), _ptr(stats), st)
            L.pp_aug_coef(_ptr(stats), None, None, 0, B, _ptr(coef), st)
            L.pp_aug_scalar_map(_ptr(x), B, H, W, _ptr(coef), _ptr(rect), st)

        norm(img, Hp, Wp, src_rect)
        clip = torch.empty_like(stats)
        L.pp_aug_stats(_ptr(img), B, Hp, Wp, _ptr(src_rect), _ptr(clip), st)
        disp = disp64 = None
        if fields is not None and fields.get('disp') is not None:
            if tuple(fields['disp'].shape) != (B, 2, Ho, Wo):
                raise ValueError(f"fields['disp'] must be {(B, 2, Ho, Wo)}, got {tuple(fields['disp'].shape)}")
            if fields['disp'].dtype == torch.float64:          # the reference's own float64 field, replayed bit for bit
                disp64 = fields['disp'].to(self.device).contiguous()
            disp = fields['disp'].to(self.device, f32).contiguous()
        elif (pk['sigma_alpha'][:, 0] > 0).any():
            disp = torch.empty(B, 2, Ho, Wo, device=self.device, dtype=f32)
            scratch = torch.empty_like(disp)
            L.pp_aug_elastic_field(_ptr(disp), _ptr(scratch), B, Ho, Wo, _ptr(self._up(pk['sigma_alpha'], f32)),
                                              pk['field_seed'], st)
        o_img = torch.empty(B, Ho, Wo, device=self.device, dtype=f32)
        o_lab = torch.empty(B, Ho, Wo, device=self.device, dtype=i32)
        o_scb = torch.empty_like(o_lab)
        valid = torch.empty_like(o_img)
        # ElasticTransform's own interpolant (scipy's cubic B-spline, augmentations.py:270) for the samples whose composed map
        # is otherwise a pure pixel permutation (mirroring / cropping): there the single resampling IS the reference's
        # map_coordinates call.  With Scaling / RandomRotation in the chain the reference resamples two or three times with
        # three libraries; those samples keep the single Keys-bicubic resampling (parity unpinned, see the module docstring).
        use = pk['spline'] if cfg.elastic_spline else np.zeros(B, np.int32)
        if disp is not None and use.any():
            spl = torch.empty(B, Hp + 24, Wp + 24, device=self.device, dtype=torch.float64)
            use_d = self._up(use, i32)
            L.pp_aug_spline_prefilter(_ptr(img), B, Hp, Wp, _ptr(maps), _ptr(use_d), _ptr(spl), st)
            L.pp_aug_warp_spline(_ptr(img), _ptr(lab), _ptr(scb), Hp, Wp, _ptr(o_img), _ptr(o_lab), _ptr(o_scb), _ptr(valid),
                                 Ho, Wo, B, _ptr(maps), _ptr(disp), _ptr(disp64), _ptr(clip), 0.0, K, 1, _ptr(spl), _ptr(use_d), st)
        else:
            L.pp_aug_warp(_ptr(img), _ptr(lab), _ptr(scb), Hp, Wp, _ptr(o_img), _ptr(o_lab), _ptr(o_scb), _ptr(valid),
                          Ho, Wo, B, _ptr(maps), _ptr(disp), _ptr(clip), 0.0, K, 1, st)
        if fields is not None and fields.get('noise') is not None:
            nz = fields['noise'].to(self.device, f32)
            if tuple(nz.shape) != (B, Ho, Wo):
                raise ValueError(f"fields['noise'] must be {(B, Ho, Wo)}, got {tuple(nz.shape)}")
            L.pp_aug_add_field(_ptr(o_img), _ptr(nz.contiguous()), B, Ho, Wo, _ptr(out_rect), st)
        elif (pk['noise'] > 0).any():
            L.pp_aug_add_noise(_ptr(o_img), B, Ho, Wo, _ptr(self._up(pk['noise'], f32)), _ptr(out_rect),
                               pk['noise_seed'], st)
        norm(o_img, Ho, Wo, out_rect)
        lab_1h = torch.empty(B, K, Ho, Wo, device=self.device, dtype=f32)
        scb_1h = torch.empty(B, K + 1, Ho, Wo, device=self.device, dtype=f32)
        L.pp_aug_onehot(_ptr(o_lab), _ptr(lab_1h), B, K, Ho * Wo, st)
        L.pp_aug_onehot(_ptr(o_scb), _ptr(scb_1h), B, K + 1, Ho * Wo, st)
        out = {'image': o_img.unsqueeze(1), 'label': lab_1h, 'scribble': scb_1h, 'valid_mask': valid.unsqueeze(1)}
        if cfg.do_strong:
            s_img = o_img.clone()
            out.update(self.strong(s_img, pk, stats, stats0, coef, raw=(image, src_rect, samples), mix=mix))
            out['label_strong'], out['scribble_strong'] = lab_1h, scb_1h
        self.last_params = dict(samples=list(samples), packed=pk)
        return out

    def strong(self, s_img, pk, stats, stats0, coef, raw=None, mix=None):
        """Brightness -> Contrast -> GammaAugmentation on the whole cropped plane (padding included, as the reference)."""
        L, B = self.lib, s_img.shape[0]
        H, W = s_img.shape[-2:]
        st = torch.cuda.current_stream(self.device).cuda_stream
        f32 = torch.float32
        bright, contrast, gamma = (self._up(pk[k], f32) for k in ('bright', 'contrast', 'gamma'))
        L.pp_aug_coef(None, None, _ptr(bright), 4, B, _ptr(coef), st)
        L.pp_aug_scalar_map(_ptr(s_img), B, H, W, _ptr(coef), None, st)
        L.pp_aug_stats(_ptr(s_img), B, H, W, None, _ptr(stats), st)
        L.pp_aug_coef(_ptr(stats), None, _ptr(contrast), 1, B, _ptr(coef), st)
        L.pp_aug_scalar_map(_ptr(s_img), B, H, W, _ptr(coef), None, st)
        L.pp_aug_stats(_ptr(s_img), B, H, W, None, _ptr(stats0), st)
        L.pp_aug_coef(_ptr(stats0), None, _ptr(gamma), 2, B, _ptr(coef), st)
        L.pp_aug_gamma(_ptr(s_img), B, H, W, _ptr(coef), None, st)
        L.pp_aug_stats(_ptr(s_img), B, H, W, None, _ptr(stats), st)
        L.pp_aug_coef(_ptr(stats), _ptr(stats0), _ptr(gamma), 3, B, _ptr(coef), st)
        L.pp_aug_scalar_map(_ptr(s_img), B, H, W, _ptr(coef), None, st)
        if (pk['blur'][:, 0] > 0).any():                                          # GaussianBlur (TransformsColorBlur)
            L.pp_aug_gaussian_blur(_ptr(s_img), _ptr(torch.empty_like(s_img)), B, H, W, _ptr(self._up(pk['blur'], f32)), st)
        if (pk['lam'] >= 0).any() and raw is not None:                            # Mixup (TransformsColorMixup)
            s_img_mix = self._partners(raw, pk, H, W, stats, coef, mix)
            L.pp_aug_mix(_ptr(s_img), _ptr(s_img_mix), B, H * W, _ptr(self._up(pk['lam'], f32)), st)
        if (pk['lowres'] > 0).any():                                              # SimulationLowRes (TransformsColorLow)
            self._lowres(s_img, pk['lowres'], stats)
        if pk['cutout'][:, 2:].prod(1).any():                                     # Cutout: image * mask (:45), mask = 0 on the square
            zero = torch.zeros(B, 4, device=self.device, dtype=f32)              # clip(0 * x + 0, -inf.., ) with bounds [0, 0]
            L.pp_aug_scalar_map(_ptr(s_img), B, H, W, _ptr(zero), _ptr(self._up(pk['cutout'], torch.int32)), st)
        return {'image_strong': s_img.unsqueeze(1)}

    def _partners(self, raw, pk, H, W, stats, coef, mix=None):
        """Mixup's second image (augmentations.py:66-71): another raw slice, centre-cropped to the canvas and normalised by its
        own mean / std.  `mix` = (planes, sizes): the loader's draw from the whole file list, one candidate per sample (the
        reference's `np.random.choice(file_ls)`); without it: slice `partner[n]` of the same batch."""
        image, src_rect, samples = raw
        L, B = self.lib, image.shape[0]
        st = torch.cuda.current_stream(self.device).cuda_stream
        f32 = torch.float32
        if mix is not None:
            src = mix[0].to(self.device, f32, non_blocking=True).contiguous()
            sizes = [(int(h), int(w)) for h, w in mix[1]]
        else:
            idx = self._up(np.where(pk['partner'] >= 0, pk['partner'], 0), None)
            src = image.to(self.device, f32, non_blocking=True)[idx].contiguous()
            own = [samples[int(pk['partner'][n])] if pk['partner'][n] >= 0 else samples[n] for n in range(B)]
            sizes = [(q['h'], q['w']) for q in own]
        Hp, Wp = src.shape[-2:]
        maps = np.zeros((B, MAP_FLOATS), np.float32)
        rect = np.zeros((B, 4), np.int32)
        for n in range(B):
            h, w = sizes[n]
            ph, pw = min(h, H), min(w, W)                                           # centre crop (:76-80); smaller slices are centred
            top, left = (H - ph) // 2, (W - pw) // 2
            maps[n] = [1, 0, h // 2 - H // 2 if h > H else -top, 0, 1, w // 2 - W // 2 if w > W else -left, top, left, ph, pw, h, w]
            rect[n] = [top, left, ph, pw]
        out = torch.zeros(B, H, W, device=self.device, dtype=f32)
        L.pp_aug_warp(_ptr(src), None, None, Hp, Wp, _ptr(out), None, None, None, H, W, B, _ptr(self._up(maps, f32)), None, None,
                      0.0, 0, 2, st)
        r = self._up(rect, torch.int32)
        L.pp_aug_stats(_ptr(out), B, H, W, _ptr(r), _ptr(stats), st)
        L.pp_aug_coef(_ptr(stats), None, None, 0, B, _ptr(coef), st)
        L.pp_aug_scalar_map(_ptr(out), B, H, W, _ptr(coef), _ptr(r), st)
        return out

    def _lowres(self, s_img, scales, stats):
        """SimulationLowRes (augmentations.py:168-182): nearest-neighbour resize to round(size / scale), cubic resize back,
        clipped to the sample's range (skimage clip=True)."""
        L, B = self.lib, s_img.shape[0]
        H, W = s_img.shape[-2:]
        st = torch.cuda.current_stream(self.device).cuda_stream
        f32 = torch.float32
        down, up = np.zeros((B, MAP_FLOATS), np.float32), np.zeros((B, MAP_FLOATS), np.float32)
        for n in range(B):
            sc = float(scales[n])
            nh, nw = (round(H / sc), round(W / sc)) if sc > 0 else (H, W)
            sy, sx = H / nh, W / nw                                                # pixel-centre convention of skimage's resize
            down[n] = [sy, 0, 0.5 * sy - 0.5, 0, sx, 0.5 * sx - 0.5, 0, 0, nh, nw, H, W]
            up[n] = [1 / sy, 0, 0.5 / sy - 0.5, 0, 1 / sx, 0.5 / sx - 0.5, 0, 0, H, W, nh, nw]
        L.pp_aug_stats(_ptr(s_img), B, H, W, None, _ptr(stats), st)
        small = torch.empty_like(s_img)
        L.pp_aug_warp(_ptr(s_img), None, None, H, W, _ptr(small), None, None, None, H, W, B, _ptr(self._up(down, f32)), None, None,
                      0.0, 0, 2, st)
        L.pp_aug_warp(_ptr(small), None, None, H, W, _ptr(s_img), None, None, None, H, W, B, _ptr(self._up(up, f32)), None,
                      _ptr(stats), 0.0, 0, 1, st)

    def __call__(self, image, label, scribble, sizes=None, mix=None, mix_sizes=None):
        """mix (B, Hm, Wm) / mix_sizes: one candidate Mixup partner slice per sample, drawn from the whole file list by the
        loader (data.NpzSlices.mix_partner) as the reference does; without it the partner is another slice of the batch."""
        if not torch.cuda.is_available():
            raise RuntimeError('DeviceAugmenter needs the GPU: the HIP library is the only implementation')
        B, Hp, Wp = image.shape
        sizes = [(Hp, Wp)] * B if sizes is None else sizes
        mixed = None
        if mix is not None:
            mixed = (mix, [(mix.shape[1], mix.shape[2])] * B if mix_sizes is None else mix_sizes)
        return self.apply(image, label, scribble, self.draw(sizes), mix=mixed)


    def ahead(self, stream, image, label, scribble, sizes=None, mix=None, mix_sizes=None):
        """The same call with the upload and the augmentation kernels on `stream` instead of the current stream, which then waits for
        them: a training loop whose host runs a step ahead of the GPU thereby executes this batch's 25 MB copy and ~1.4 ms of
        kernels beside the previous training step instead of in front of this one.  stream None: plain call."""
        if stream is None:
            return self(image, label, scribble, sizes, mix, mix_sizes)
        with torch.cuda.stream(stream):
            out = self(image, label, scribble, sizes, mix, mix_sizes)
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(stream)
        for v in out.values():
            if torch.is_tensor(v):
                v.record_stream(main)
        return out


def collate_raw(items):
    """DataLoader collate for un-augmented slices ({'img', 'lab', 'scb'} numpy arrays): zero-pad to one plane size."""
    hs, ws = [it['img'].shape[0] for it in items], [it['img'].shape[1] for it in items]
    Hp, Wp = max(hs), max(ws)
    B = len(items)
    img = np.zeros((B, Hp, Wp), np.float32)
    lab = np.zeros((B, Hp, Wp), np.int32)
    scb = np.zeros((B, Hp, Wp), np.int32)
    for i, it in enumerate(items):
        img[i, :hs[i], :ws[i]], lab[i, :hs[i], :ws[i]], scb[i, :hs[i], :ws[i]] = it['img'], it['lab'], it['scb']
    out = dict(img=torch.from_numpy(img), lab=torch.from_numpy(lab), scb=torch.from_numpy(scb), sizes=list(zip(hs, ws)))
    if 'mix' in items[0]:                      # Mixup partner candidates from the whole file list (data.NpzSlices.mix_partner)
        mh, mw = [it['mix'].shape[0] for it in items], [it['mix'].shape[1] for it in items]
        mix = np.zeros((B, max(mh), max(mw)), np.float32)
        for i, it in enumerate(items):
            mix[i, :mh[i], :mw[i]] = it['mix']
        out['mix'], out['mix_sizes'] = torch.from_numpy(mix), list(zip(mh, mw))
    return out
