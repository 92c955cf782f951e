"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the input pipeline; nothing under pacingpseudo_amd/
may import this file.

Two layers:
  (A) the reference's transforms that involve no interpolation, restated line by line in numpy from
      /root/reference/datasets/augmentations.py (MeanStdNorm :11-21, Brightness :97-110, Contrast :112-129,
      GammaAugmentation :131-166, Mirroring :337-351, GaussianNoise :353-366 given the normal field, RandomCrop :368-418
      given the drawn offsets, to_one_hot_encoding :448-461) and the displacement field of ElasticTransform :259-260 given
      the uniform fields (scipy.ndimage.gaussian_filter -- scipy IS the reference's implementation there);
  (B) the definitions the device kernels add on top: the Philox-4x32-10 counter RNG, the single bicubic (Keys a = -0.75)
      / nearest resampling of the composed affine map, and the batch pipeline that strings (A) and (B) together in the
      order pacingpseudo_amd/augment.py launches them.

PARITY STATUS (round 3): layer (A) is PINNED by vectors captured from the reference module itself, imported in the build
container with empty placeholder modules for cv2 / skimage (tests/golden/make_golden_r3.py -> tests/golden/aug_ref.npz,
checked by tests/test_oracle_golden_r3.py): every transform whose own code is numpy / scipy only, and whole samples of the
reference's CHAOSTwoStream dataset class under the CHAOS recipe.  Scaling and SimulationLowRes (skimage.transform.resize)
and RandomRotation (cv2.warpAffine) cannot run without those libraries and are NOT restated: "parity unpinned" for them.
"""
import numpy as np
import scipy.ndimage

SKIP = -1.0e30
EPS = np.float32(1e-8)


# ------------------------------------------------------------------------------------------ (A) reference arithmetic
def mean_std_norm(image):                      # augmentations.py:16-21
    return (image - np.mean(image)) / (np.std(image) + 1e-8)


def brightness(image, scale):                  # :107-109
    return image + scale


def contrast(image, scale):                    # :122-128
    mean_, max_, min_ = np.mean(image), np.max(image), np.min(image)
    return np.clip((image - mean_) * scale + mean_, min_, max_)


def gamma_augmentation(image, gamma):          # :145-165 (retain_stats=True, invert_data=False)
    mean_, std_, max_, min_ = np.mean(image), np.std(image), np.max(image), np.min(image)
    image = np.power((image - min_) / (max_ - min_ + 1e-8), gamma)
    image = (image - np.mean(image)) / (np.std(image) + 1e-8)
    return image * std_ + mean_


def mirroring(arrs, axis):                     # :346-350
    return [np.flip(a, axis) for a in arrs]


def random_crop(image, label, scb, crop_size, image_top, image_left, canvas_top, canvas_left, image_pad=0, label_pad=4):
    """:379-417 with the four offsets already drawn."""
    h, w = image.shape
    crop_h, crop_w = crop_size
    patch_w, patch_h = min(w, crop_w), min(h, crop_h)
    out = []
    for a, pad in ((image, image_pad), (label, label_pad), (scb, label_pad)):
        canvas = np.zeros(crop_size, np.float32) + pad
        canvas[canvas_top:canvas_top + patch_h, canvas_left:canvas_left + patch_w] = \
            a[image_top:image_top + patch_h, image_left:image_left + patch_w]
        out.append(canvas)
    valid = np.zeros(crop_size, np.float32)
    valid[canvas_top:canvas_top + patch_h, canvas_left:canvas_left + patch_w] = 1.
    return out[0], out[1], out[2], valid


def gaussian_blur(image, scale):               # :93-94
    return scipy.ndimage.gaussian_filter(image, scale, order=0)


def mixup(image1, image2_raw, lam):            # :66-72 (image2 already centre-cropped to image1's shape)
    image2 = (image2_raw - image2_raw.mean()) / max(image2_raw.std(), 1e-8)
    return image1 * lam + image2 * (1 - lam)


def center_crop(image, h, w):                  # :75-80
    h0, w0 = image.shape
    y, x = h0 // 2, w0 // 2
    return image[y - h // 2: y + h // 2, x - w // 2: x + w // 2]


def to_one_hot(image, n):                      # :448-461
    out = np.zeros((n,) + image.shape, np.float32)
    for c in range(n):
        out[c][image == c] = 1
    return out


def elastic_field(uniform, sigma, alpha):      # :259-260, `uniform` = np.random.rand(h, w) * 2 - 1
    return scipy.ndimage.gaussian_filter(uniform, sigma) * alpha


# ---- scipy.ndimage.map_coordinates(order=3, mode='nearest') restated (the interpolant of ElasticTransform :270) ----
# The arithmetic lives in scipy (the reference pins no version, README.md:42 era ~1.5; the build image has 1.15.3), not in
# the reference; its published algorithm (scipy/ndimage/_interpolation.py map_coordinates + src/ni_splines.c, ni_interpolation.c):
#   1. the input is padded by 12 pixels per side with its edge values (_prepad_for_spline_filter, mode 'nearest');
#   2. cubic B-spline prefilter along axis 0, then axis 1, in float64: gain (1 - z)(1 - 1/z), z = sqrt(3) - 2, causal and
#      anticausal recursion with the 'reflect' initialisation (the one scipy uses for mode 'nearest');
#   3. per output point: coordinate + 12, clamped to the padded array, four taps from floor(c) - 1 with the cubic
#      B-spline weights of get_spline_interpolation_weights, tap indices clamped to the padded array; result cast to the
#      input dtype.
# tests/test_augment.py checks this restatement against scipy itself; the device kernels (pp_aug_spline_prefilter,
# pp_aug_warp_spline) follow it line by line.
SPLINE_PAD = 12
SPLINE_POLE = np.sqrt(3.0) - 2.0


def spline_prefilter_line(c):
    """In-place cubic B-spline prefilter of one float64 line, 'reflect' initialisation (ni_splines.c: _init_causal_reflect,
    _init_anticausal_reflect)."""
    z, n = SPLINE_POLE, len(c)
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    z_i, z_n, c0 = z, z ** n, c[0]
    c[0] = c[0] + z_n * c[n - 1]
    for i in range(1, n):
        c[0] += z_i * (c[i] + z_n * c[n - 1 - i])
        z_i *= z
    c[0] *= z / (1.0 - z_n * z_n)
    c[0] += c0
    for i in range(1, n):
        c[i] += z * c[i - 1]
    c[n - 1] *= z / (z - 1.0)
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return c


def spline_coefficients(image):
    """(h + 24, w + 24) float64 cubic B-spline coefficients of an image: steps 1-2 above."""
    c = np.pad(np.asarray(image, np.float64), SPLINE_PAD, mode='edge')
    # axis 0 first (scipy filters the axes in order)
    for col in range(c.shape[1]):
        c[:, col] = spline_prefilter_line(c[:, col].copy())
    for r in range(c.shape[0]):
        c[r, :] = spline_prefilter_line(c[r, :].copy())
    return c


def spline_weights(t):
    """get_spline_interpolation_weights, order 3: taps at floor(c) - 1 .. floor(c) + 2, t = c - floor(c)."""
    zz = 1.0 - t
    w1 = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0
    w2 = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0
    w0 = zz * zz * zz / 6.0
    return w0, w1, w2, 1.0 - w0 - w1 - w2


def map_coordinates_cubic_nearest(image, ys, xs):
    """scipy.ndimage.map_coordinates(image, (ys, xs), order=3, mode='nearest') restated; ys / xs any shape, float64."""
    coef = spline_coefficients(image)
    Hq, Wq = coef.shape
    cy = np.clip(np.asarray(ys, np.float64) + SPLINE_PAD, 0.0, Hq - 1.0)
    cx = np.clip(np.asarray(xs, np.float64) + SPLINE_PAD, 0.0, Wq - 1.0)
    fy, fx = np.floor(cy), np.floor(cx)
    wy, wx = spline_weights(cy - fy), spline_weights(cx - fx)
    out = np.zeros(cy.shape, np.float64)
    for r in range(4):
        yy = np.clip(fy.astype(np.int64) - 1 + r, 0, Hq - 1)
        row = np.zeros(cy.shape, np.float64)
        for c in range(4):
            xx = np.clip(fx.astype(np.int64) - 1 + c, 0, Wq - 1)
            row += wx[c] * coef[yy, xx]
        out += wy[r] * row
    return out.astype(np.asarray(image).dtype)


def rotation90(a, num_rots, axes=(0, 1)):      # :330-333
    return np.rot90(a, num_rots, axes=axes)


def cutout(image, length, y, x):               # :34-46, centre (y, x) already drawn
    h, w = image.shape
    mask = np.ones((h, w), np.float32)
    y1, y2 = np.clip(y - length // 2, 0, h), np.clip(y + length // 2, 0, h)
    x1, x2 = np.clip(x - length // 2, 0, w), np.clip(x + length // 2, 0, w)
    mask[y1:y2, x1:x2] = 0.
    return image * mask


def elastic_apply(image, label, scb, dx, dy, img_order=3, lab_order=0, mode='nearest', clip=True):
    """:262-271 given the two displacement fields: cubic-spline map_coordinates for the image (clipped to its range),
    order 0 for the class maps; scipy.ndimage IS the reference's implementation here."""
    h, w = image.shape
    min_, max_ = image.min(), image.max()
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    x_dx, y_dy = np.reshape(x + dx, (-1, 1)), np.reshape(y + dy, (-1, 1))
    out = scipy.ndimage.map_coordinates(image, (y_dy, x_dx), order=img_order, mode=mode).reshape(h, w)
    if clip:
        out = np.clip(out, min_, max_)
    lab = scipy.ndimage.map_coordinates(label, (y_dy, x_dx), order=lab_order, mode=mode).reshape(h, w)
    s = scipy.ndimage.map_coordinates(scb, (y_dy, x_dx), order=lab_order, mode=mode).reshape(h, w)
    return out, lab, s


def reference_two_stream(raw, draws, arrays, crop_size, K, strength=1.0):
    """One sample of CHAOSTwoStream.__getitem__ (datasets/chaos/chaos_dataset.py:58-105) under TransformsColor
    (chaos_aug_configs.py:16-86) for a draw sequence in which Scaling and RandomRotation do not fire.  `draws`: the
    scalar results of the numpy.random calls in call order, `arrays`: the array-valued ones (np.random.rand fields of
    ElasticTransform, the np.random.normal field of GaussianNoise).  Returns (dict like the reference's, set of the
    transforms that fired)."""
    d, arrays, fired = [float(x) for x in draws], list(arrays), set()
    img, lab, scb = (a.astype(np.float32) for a in raw)
    img = mean_std_norm(img)
    assert not d.pop(0) < 0.2, 'Scaling fired: unpinned transform'
    if d.pop(0) < 0.2:                                   # ElasticTransform
        sigma, alpha = d.pop(0), d.pop(0)
        dx = elastic_field(arrays.pop(0) * 2 - 1, sigma, alpha)
        dy = elastic_field(arrays.pop(0) * 2 - 1, sigma, alpha)
        img, lab, scb = elastic_apply(img, lab, scb, dx, dy)
        fired.add('elastic')
    assert not d.pop(0) < 0.2, 'RandomRotation fired: unpinned transform'
    for axis in (0, 1):
        if d.pop(0) < 0.5:
            img, lab, scb = mirroring([img, lab, scb], axis)
            fired.add(f'mirror{axis}')
    if d.pop(0) < 0.15:                                  # GaussianNoise :360-365
        d.pop(0)
        img = img + arrays.pop(0)
        fired.add('noise')
    img = mean_std_norm(img)
    assert d.pop(0) < 1.0                                # RandomCrop gate (p = 1)
    h, w = img.shape
    first, second = int(d.pop(0)), int(d.pop(0))         # width offset first, then height (:386-397)
    il, cl = (first, 0) if w - crop_size[1] > 0 else (0, first)
    it, ct = (second, 0) if h - crop_size[0] > 0 else (0, second)
    img, lab, scb, valid = random_crop(img, lab, scb, crop_size, it, il, ct, cl, 0, K)
    out = dict(image=img[None], label=to_one_hot(lab, K), scribble=to_one_hot(scb, K + 1), valid_mask=valid[None])
    s = img
    if d.pop(0) < 0.8:
        s = brightness(s, d.pop(0)); fired.add('brightness')
    if d.pop(0) < 0.8:
        s = contrast(s, d.pop(0)); fired.add('contrast')
    if d.pop(0) < 0.8:
        d.pop(0)                                         # the gamma < 1 coin (:151)
        s = gamma_augmentation(s, d.pop(0)); fired.add('gamma')
    assert not d and not arrays, 'draws left over'
    out.update(image_strong=s[None], label_strong=out['label'], scribble_strong=out['scribble'])
    return out, fired


# ------------------------------------------------------------------------------------------ (B) device definitions
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11), vectorised over counters."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    m = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m
        n1 = p1 & m
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m
        n3 = p0 & m
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & m
        k1 = (k1 + np.uint64(0xBB67AE85)) & m
    return c0, c1, c2, c3


def uniform_field(total, seed):
    """aug_uniform_kernel: U(-1, 1) from the top 24 bits of each Philox word, counter = element quad."""
    q = np.arange((total + 3) // 4, dtype=np.uint64)
    r = philox4x32_10(q & 0xFFFFFFFF, q >> np.uint64(32), 0x5eed, 0, seed & 0xFFFFFFFF, seed >> 32)
    w = np.stack(r, 1).reshape(-1)[:total]
    return (((w >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def normal_field(B, HW, seed):
    """aug_noise_kernel: Box-Muller on Philox words, counter = (pixel quad, sample); returns (B, HW) float32 N(0, 1)."""
    quads = (HW + 3) // 4
    out = np.zeros((B, quads * 4), np.float32)
    q = np.arange(quads, dtype=np.uint64)
    for n in range(B):
        r = philox4x32_10(q, 0, n, 0, seed & 0xFFFFFFFF, seed >> 32)
        u = [((x >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24) for x in r]
        z = np.zeros((quads, 4), np.float32)
        for h in range(2):
            rad = np.sqrt(np.float32(-2.0) * np.log(u[2 * h]))
            ang = np.float32(6.283185307179586) * u[2 * h + 1]
            z[:, 2 * h], z[:, 2 * h + 1] = rad * np.cos(ang), rad * np.sin(ang)
        out[n] = z.reshape(-1)
    return out[:, :HW]


def device_elastic_field(B, H, W, sigma_alpha, seed):
    """pp_aug_elastic_field: [B][2][H][W]; scipy's gaussian_filter (truncate 4, 'reflect') IS the separable filter used."""
    u = uniform_field(B * 2 * H * W, seed).reshape(B, 2, H, W)
    out = np.zeros_like(u)
    for n in range(B):
        sg, al = float(sigma_alpha[n][0]), float(sigma_alpha[n][1])
        if sg > 0:
            for a in range(2):
                out[n, a] = scipy.ndimage.gaussian_filter(u[n, a].astype(np.float64), sg) * al
    return out.astype(np.float32)


def stats(x, rect=None):
    """pp_aug_stats for one sample: mean, std, min, max over the rectangle (float64 accumulation of float32 values)."""
    if rect is not None:
        t, l, h, w = rect
        x = x[t:t + h, l:l + w]
    x64 = x.astype(np.float64)
    mean = x64.mean() if x.size else 0.0
    var = max((x64 * x64).mean() - mean * mean, 0.0) if x.size else 0.0
    return np.array([mean, np.sqrt(var), x.min() if x.size else 0.0, x.max() if x.size else 0.0], np.float64)


def coef(mode, st=None, st0=None, param=None):
    """pp_aug_coef for one sample (float32 arithmetic as on the device)."""
    f = np.float32
    inf = f(3.0e38)
    c = np.array([1, 0, -inf, inf], np.float32)
    if mode == 2:
        c = np.array([0, 1, -1, 0], np.float32)
    if param is not None and param <= SKIP:
        return c
    if mode == 0:
        a = f(1) / (f(st[1]) + EPS)
        c[0], c[1] = a, -f(st[0]) * a
    elif mode == 1:
        c[:] = [f(param), f(st[0]) * (f(1) - f(param)), f(st[2]), f(st[3])]
    elif mode == 2:
        c[:] = [f(st[2]), f(st[3]) - f(st[2]) + EPS, f(param), 0]
    elif mode == 3:
        a = f(st0[1]) / (f(st[1]) + EPS)
        c[0], c[1] = a, f(st0[0]) - f(st[0]) * a
    else:
        c[1] = f(param)
    return c


def scalar_map(x, c, rect=None):
    y = x.copy()
    sl = (slice(None), slice(None)) if rect is None else (slice(rect[0], rect[0] + rect[2]), slice(rect[1], rect[1] + rect[3]))
    y[sl] = np.minimum(np.maximum(c[0] * x[sl] + c[1], c[2]), c[3])
    return y


def gamma_map(x, c):
    if c[2] <= 0:
        return x.copy()
    return np.power(np.maximum((x - c[0]) / c[1], np.float32(0)), c[2]).astype(np.float32)


def keys_weights(t):
    a = np.float32(-0.75)
    t = t.astype(np.float32)
    one = np.float32(1)
    w0 = ((a * (t + one) - np.float32(5) * a) * (t + one) + np.float32(8) * a) * (t + one) - np.float32(4) * a
    w1 = ((a + np.float32(2)) * t - (a + np.float32(3))) * t * t + one
    u = one - t
    w2 = ((a + np.float32(2)) * u - (a + np.float32(3))) * u * u + one
    return [w0, w1, w2, one - w0 - w1 - w2]


def warp(img, lab, scb, m, Ho, Wo, disp=None, clip=None, img_pad=0.0, lab_pad=4, cubic=True, spline=False):
    """aug_warp_kernel for one sample.  img / lab / scb: (Hp, Wp) planes; m: the 12 map floats.  spline: the image through
    scipy's cubic B-spline (map_coordinates_cubic_nearest above) and the class maps rounded from double coordinates -- the
    path of a sample whose only interpolating transform is ElasticTransform (pp_aug_warp_spline)."""
    f = np.float32
    m = np.asarray(m, np.float32)
    top, left, ph, pw, hs, ws = (int(v) for v in m[6:12])
    yo, xo = np.meshgrid(np.arange(Ho, dtype=np.float32), np.arange(Wo, dtype=np.float32), indexing='ij')
    valid = (yo >= top) & (yo < top + ph) & (xo >= left) & (xo < left + pw)
    ys = (m[0] * yo + m[1] * xo + m[2]).astype(f)
    xs = (m[3] * yo + m[4] * xo + m[5]).astype(f)
    if disp is not None:
        inside = (ys >= -0.5) & (ys < hs - 0.5) & (xs >= -0.5) & (xs < ws - 0.5)
        ys2, xs2 = (ys + disp[0]).astype(f), (xs + disp[1]).astype(f)
        ys = np.where(inside, np.clip(ys2, 0, hs - 1), ys2).astype(f)
        xs = np.where(inside, np.clip(xs2, 0, ws - 1), xs2).astype(f)
    yn, xn = np.floor(ys + f(0.5)).astype(np.int64), np.floor(xs + f(0.5)).astype(np.int64)
    if spline:
        d64 = np.float64
        yd = m[0].astype(d64) * yo + m[1].astype(d64) * xo + m[2].astype(d64)
        xd = m[3].astype(d64) * yo + m[4].astype(d64) * xo + m[5].astype(d64)
        if disp is not None:
            yd, xd = yd + np.asarray(disp[0], d64), xd + np.asarray(disp[1], d64)
        yu, xu = yd, xd
        if disp is not None:
            yd = np.where(inside, np.clip(yd, 0, hs - 1), yd)
            xd = np.where(inside, np.clip(xd, 0, ws - 1), xd)
        yn, xn = np.floor(yd + 0.5).astype(np.int64), np.floor(xd + 0.5).astype(np.int64)
    in_src = (yn >= 0) & (yn < hs) & (xn >= 0) & (xn < ws)
    ync, xnc = np.clip(yn, 0, hs - 1), np.clip(xn, 0, ws - 1)
    o_lab = np.where(valid & in_src, lab[ync, xnc], lab_pad).astype(np.int32)
    o_scb = np.where(valid & in_src, scb[ync, xnc], lab_pad).astype(np.int32)
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < hs) & (xx >= 0) & (xx < ws)
        return np.where(ok, img[np.clip(yy, 0, hs - 1), np.clip(xx, 0, ws - 1)], f(img_pad)).astype(f)
    if spline:
        v = map_coordinates_cubic_nearest(img[:hs, :ws].astype(np.float64), yu, xu).astype(f)
    elif cubic == 2:
        v = tap(yn, xn)
    elif cubic:
        wy, wx = keys_weights(ys - y0.astype(f)), keys_weights(xs - x0.astype(f))
        v = np.zeros((Ho, Wo), f)
        for r in range(4):
            row = np.zeros((Ho, Wo), f)
            for c in range(4):
                row = row + wx[c] * tap(y0 - 1 + r, x0 - 1 + c)
            v = v + wy[r] * row
    else:
        ty, tx = ys - y0.astype(f), xs - x0.astype(f)
        v = (1 - ty) * ((1 - tx) * tap(y0, x0) + tx * tap(y0, x0 + 1)) + ty * ((1 - tx) * tap(y0 + 1, x0) + tx * tap(y0 + 1, x0 + 1))
    if clip is not None:
        v = np.minimum(np.maximum(v, min(f(clip[2]), f(img_pad))), max(f(clip[3]), f(img_pad)))
    v = np.where(valid & in_src, v, f(img_pad)).astype(f)
    return v, o_lab, o_scb, valid.astype(f)


def pipeline(image, label, scribble, packed, crop_size, K, do_strong=True):
    """The batch pipeline of pacingpseudo_amd.augment.DeviceAugmenter.apply, sample by sample in numpy."""
    B = image.shape[0]
    Ho, Wo = crop_size
    disp = None
    if (packed['sigma_alpha'][:, 0] > 0).any():
        disp = device_elastic_field(B, Ho, Wo, packed['sigma_alpha'], packed['field_seed'])
    nz = normal_field(B, Ho * Wo, packed['noise_seed']).reshape(B, Ho, Wo) if (packed['noise'] > 0).any() else None
    out = dict(image=[], label=[], scribble=[], valid_mask=[], image_strong=[])
    for n in range(B):
        sr, orc = packed['src_rect'][n], packed['out_rect'][n]
        img = image[n].astype(np.float32)
        img = scalar_map(img, coef(0, stats(img, sr)), sr)
        clip = stats(img, sr)
        v, ol, os_, valid = warp(img, label[n], scribble[n], packed['maps'][n], Ho, Wo, None if disp is None else disp[n],
                                 clip, 0.0, K, True, spline=bool(disp is not None and packed.get('spline', np.zeros(B))[n]))
        if packed['noise'][n] > 0:
            t, l, h, w = orc
            v[t:t + h, l:l + w] += packed['noise'][n] * nz[n, t:t + h, l:l + w]
        v = scalar_map(v, coef(0, stats(v, orc)), orc)
        out['image'].append(v[None]); out['valid_mask'].append(valid[None])
        out['label'].append(to_one_hot(ol, K)); out['scribble'].append(to_one_hot(os_, K + 1))
        if do_strong:
            s = scalar_map(v, coef(4, param=packed['bright'][n]))
            s = scalar_map(s, coef(1, stats(s), param=packed['contrast'][n]))
            st0 = stats(s)
            s = gamma_map(s, coef(2, st0, param=packed['gamma'][n]))
            s = scalar_map(s, coef(3, stats(s), st0, param=packed['gamma'][n]))
            out['image_strong'].append(s[None])
    return {k: np.stack(v) for k, v in out.items() if v}


def lowres(image, scale, clip_stats):
    """DeviceAugmenter._lowres for one (H, W) plane: nearest down to round(size / scale), Keys-cubic back up, clipped."""
    H, W = image.shape
    nh, nw = round(H / scale), round(W / scale)
    sy, sx = H / nh, W / nw
    z = np.zeros_like(image, dtype=np.int32)
    down = np.array([sy, 0, 0.5 * sy - 0.5, 0, sx, 0.5 * sx - 0.5, 0, 0, nh, nw, H, W], np.float32)
    up = np.array([1 / sy, 0, 0.5 / sy - 0.5, 0, 1 / sx, 0.5 / sx - 0.5, 0, 0, H, W, nh, nw], np.float32)
    small = warp(image, z, z, down, H, W, None, None, 0.0, 0, 2)[0]
    return warp(small, z, z, up, H, W, None, clip_stats, 0.0, 0, 1)[0]
