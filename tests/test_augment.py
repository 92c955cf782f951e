"""CPU tests of the input-pipeline host logic and of the oracle for it (no GPU): the Philox generator against the
Random123 known-answer vectors, the composed affine map against the reference's flip / crop arithmetic, and the order in
which draw_sample consumes the random stream against the order of datasets/augmentations.py."""
import numpy as np
import pytest

from oracle import augment_oracle as AO
from pacingpseudo_amd import augment as A


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        assert tuple(int(x) for x in AO.philox4x32_10(*c, *k)) == want


def test_noise_fields_have_the_right_law():
    u = AO.uniform_field(1 << 16, 1234)
    assert u.min() > -1 and u.max() < 1 and abs(u.mean()) < 0.01 and abs(u.std() - 1 / np.sqrt(3)) < 0.01
    z = AO.normal_field(2, 1 << 15, 99)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and not np.array_equal(z[0], z[1])


class _Recorder:
    """RandomState that logs which draw was made (the reference uses the same legacy numpy functions)."""

    def __init__(self, seed):
        self.rs, self.log = np.random.RandomState(seed), []

    def uniform(self, *a):
        self.log.append(('uniform',) + tuple(round(float(x), 6) for x in a))
        return self.rs.uniform(*a)

    def randint(self, n):
        self.log.append(('randint', int(n)))
        return self.rs.randint(n)


def _expected_calls(p, cfg):
    """The draws datasets/augmentations.py makes for these outcomes, transform by transform (chaos_aug_configs.py:16-86)."""
    e = [('uniform',)]                                                   # Scaling gate :200
    if p['scale'] is not None:
        e.append(('uniform',) + cfg.scale_range)                         # :204
    e.append(('uniform',))                                               # ElasticTransform gate :248
    if p['sigma'] > 0:
        e += [('uniform',) + cfg.sigma_range, ('uniform',) + cfg.alpha_range, ('randint', 2 ** 31 - 1)]   # :256-257 (+ device seed)
    e.append(('uniform',))                                               # RandomRotation gate :299
    if p['degree'] is not None:
        e.append(('uniform',) + cfg.degree_range)                        # :304
    e += [('uniform',), ('uniform',), ('uniform',)]                      # Mirroring x2 :343, GaussianNoise gate :360
    if p['noise'] > 0:
        e += [('uniform',) + cfg.noise_scale_range, ('randint', 2 ** 31 - 1)]                              # :364 (+ device seed)
    e.append(('uniform',))                                               # RandomCrop gate :377
    e.append(('randint', abs(p['nw'] - cfg.crop_size[1]) + 1))           # :387 / :391
    e.append(('randint', abs(p['nh'] - cfg.crop_size[0]) + 1))           # :393 / :397
    for key, rng in (('bright', (-0.8, 0.8)), ('contrast', (0.2, 1.8))):  # :103-107, :120-124
        e.append(('uniform',))
        if p[key] > A.SKIP:
            e.append(('uniform',) + rng)
    e.append(('uniform',))                                               # GammaAugmentation gate :141
    if p['gamma'] > A.SKIP:
        e.append(('uniform',))                                           # :151
        e.append(('uniform', 0.2, 1.0) if p['gamma'] < 1.0 else ('uniform', 1.0, 1.8))
    return [tuple(round(float(x), 6) if not isinstance(x, str) and i and t[0] == 'uniform' else x for i, x in enumerate(t)) for t in e]


def test_draw_order_follows_the_reference_transform_list():
    cfg = A.AugConfig()
    seen = set()
    for seed in range(200):
        r = _Recorder(seed)
        p = A.draw_sample(r, 200 + seed % 100, 180 + seed % 130, cfg)
        assert r.log == _expected_calls(p, cfg), seed
        seen |= {k for k in ('scale', 'degree') if p[k] is not None} | {k for k in ('sigma', 'noise') if p[k] > 0}
        assert p['patch_h'] == min(p['nh'], 256) and p['patch_w'] == min(p['nw'], 256)
        assert 0 <= p['canvas_top'] <= 256 - p['patch_h'] and 0 <= p['image_top'] <= p['nh'] - p['patch_h']
    assert seen == {'scale', 'degree', 'sigma', 'noise'}


@pytest.mark.parametrize('h,w', [(256, 256), (200, 300), (300, 210), (320, 330), (180, 150)])
@pytest.mark.parametrize('flip0,flip1', [(False, False), (True, False), (False, True), (True, True)])
def test_composed_map_equals_flip_then_crop(h, w, flip0, flip1):
    """Without Scaling / rotation the single resampling must reproduce np.flip + RandomCrop exactly (augmentations.py:346-417)."""
    rng = np.random.RandomState(h * 7 + w + flip0 * 2 + flip1)
    cfg = A.AugConfig(p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0)
    p = A.draw_sample(rng, h, w, cfg)
    p['flip0'], p['flip1'] = flip0, flip1
    img = rng.normal(size=(h, w)).astype(np.float32)
    lab = rng.randint(0, 5, (h, w)).astype(np.int32)
    scb = rng.randint(0, 6, (h, w)).astype(np.int32)
    ri, rl, rs = img, lab, scb
    if flip0:
        ri, rl, rs = AO.mirroring([ri, rl, rs], 0)
    if flip1:
        ri, rl, rs = AO.mirroring([ri, rl, rs], 1)
    ri, rl, rs, rv = AO.random_crop(ri, rl, rs, cfg.crop_size, p['image_top'], p['image_left'], p['canvas_top'],
                                    p['canvas_left'], 0, 5)
    v, ol, os_, valid = AO.warp(img, lab, scb, A.compose_map(p), 256, 256, None, None, 0.0, 5, True)
    np.testing.assert_array_equal(valid, rv)
    np.testing.assert_array_equal(ol, rl.astype(np.int32))
    np.testing.assert_array_equal(os_, rs.astype(np.int32))
    np.testing.assert_allclose(v, ri, rtol=0, atol=1e-6)          # Keys weights at t = 0 are (0, 1, 0, 0) up to rounding


def test_scaling_map_samples_pixel_centres_unpinned():
    p = A.draw_sample(np.random.RandomState(0), 100, 100, A.AugConfig(p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0))
    p.update(scale=2.0, nh=200, nw=200, flip0=False, flip1=False, image_top=0, image_left=0, canvas_top=0, canvas_left=0,
             patch_h=200, patch_w=200)
    m = A.compose_map(p)
    # skimage.transform.resize convention: output pixel i looks at (i + 0.5) / 2 - 0.5
    assert np.allclose([m[0], m[2], m[4], m[5]], [0.5, -0.25, 0.5, -0.25])


def test_collate_raw_pads_to_one_plane():
    items = [dict(img=np.ones((4, 6), np.float32), lab=np.ones((4, 6), np.int32), scb=np.ones((4, 6), np.int32)),
             dict(img=np.ones((5, 3), np.float32), lab=np.ones((5, 3), np.int32), scb=np.ones((5, 3), np.int32))]
    b = A.collate_raw(items)
    assert b['img'].shape == (2, 5, 6) and b['sizes'] == [(4, 6), (5, 3)]
    assert float(b['img'][1, :, 3:].abs().sum()) == 0 and float(b['img'][0, 4].abs().sum()) == 0


def test_reference_arithmetic_vs_device_definitions_on_cpu():
    """Layer (B) of the oracle (float32, device order of operations) against layer (A) (the reference's numpy lines)."""
    rng = np.random.RandomState(3)
    x = (rng.normal(size=(64, 48)) * 40 + 100).astype(np.float32)
    n = AO.scalar_map(x, AO.coef(0, AO.stats(x)))
    np.testing.assert_allclose(n, AO.mean_std_norm(x.astype(np.float64)), atol=2e-5)
    s = AO.scalar_map(n, AO.coef(4, param=np.float32(0.3)))
    np.testing.assert_allclose(s, AO.brightness(n.astype(np.float64), np.float32(0.3)), atol=1e-6)
    c = AO.scalar_map(s, AO.coef(1, AO.stats(s), param=np.float32(1.4)))
    np.testing.assert_allclose(c, AO.contrast(s.astype(np.float64), np.float32(1.4)), atol=2e-5)
    st0 = AO.stats(c)
    g = AO.gamma_map(c, AO.coef(2, st0, param=np.float32(0.7)))
    g = AO.scalar_map(g, AO.coef(3, AO.stats(g), st0, param=np.float32(0.7)))
    np.testing.assert_allclose(g, AO.gamma_augmentation(c.astype(np.float64), np.float32(0.7)), atol=5e-5)


@pytest.mark.parametrize('k', [1, 2, 3])
@pytest.mark.parametrize('h,w', [(40, 56), (33, 21)])
def test_rotation90_map_is_np_rot90(k, h, w):
    """Rotation90 (augmentations.py:319-335) folded into the composed map: exactly np.rot90(., k, axes=(0, 1))."""
    rng = np.random.RandomState(k * 100 + h)
    oh, ow = (w, h) if k % 2 else (h, w)
    cfg = A.AugConfig(crop_size=(oh, ow), p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, p_mirror=0, p_rot90=1.0, do_strong=False)
    log = _Recorder(5)
    p = A.draw_sample(log, h, w, cfg)
    assert ('randint', 3) in log.log and p['rot90'] in (1, 2, 3)
    p['rot90'] = k
    p['nh'], p['nw'] = oh, ow
    p.update(image_top=0, image_left=0, canvas_top=0, canvas_left=0, patch_h=oh, patch_w=ow)
    img = rng.normal(size=(h, w)).astype(np.float32)
    lab = rng.randint(0, 5, (h, w)).astype(np.int32)
    v, ol, os_, valid = AO.warp(img, lab, lab, A.compose_map(p), oh, ow, None, None, 0.0, 5, True)
    assert valid.all()
    np.testing.assert_array_equal(ol, AO.rotation90(lab, k))
    np.testing.assert_allclose(v, AO.rotation90(img, k), atol=1e-6)


def test_cutout_rect_is_the_reference_square():
    for (y, x, h, w) in [(0, 0, 40, 50), (39, 49, 40, 50), (20, 3, 40, 50), (7, 25, 16, 30)]:
        t, l, hh, ww = A.cutout_rect(y, x, 16, h, w)
        m = np.ones((h, w), np.float32)
        m[t:t + hh, l:l + ww] = 0
        np.testing.assert_array_equal(m, AO.cutout(np.ones((h, w), np.float32), 16, y, x))


def test_cubic_spline_restatement_against_scipy():
    """oracle/augment_oracle.py restates scipy.ndimage.map_coordinates(order=3, mode='nearest') -- the interpolant of the
    reference's ElasticTransform (datasets/augmentations.py:270); the device kernels follow the restatement.  scipy is
    installed in the build image, so the restatement is checked against scipy itself: coefficients and interpolated values to
    1e-13 wherever the coordinate lies within the 12-pixel padding scipy adds (elastic displacements are a few pixels), and to
    1e-6 beyond it."""
    import scipy.ndimage as ndi
    from oracle import augment_oracle as AO
    rng = np.random.RandomState(3)
    for shape in ((37, 45), (64, 64), (5, 9)):
        img = rng.normal(size=shape)
        coef = AO.spline_coefficients(img)
        ref = ndi.spline_filter(np.pad(img, AO.SPLINE_PAD, mode='edge'), 3, mode='nearest')
        assert np.abs(coef - ref).max() < 1e-13
        ys = rng.uniform(-20, shape[0] + 20, 3000)
        xs = rng.uniform(-20, shape[1] + 20, 3000)
        want = ndi.map_coordinates(img, (ys, xs), order=3, mode='nearest')
        got = AO.map_coordinates_cubic_nearest(img, ys, xs)
        near = (ys >= -11) & (ys <= shape[0] + 10) & (xs >= -11) & (xs <= shape[1] + 10)
        assert np.abs(got - want)[near].max() < 1e-13
        assert np.abs(got - want).max() < 1e-6
        img32 = img.astype(np.float32)
        got32 = AO.map_coordinates_cubic_nearest(img32, ys, xs)
        assert got32.dtype == np.float32
        assert np.abs(got32 - ndi.map_coordinates(img32, (ys, xs), order=3, mode='nearest'))[near].max() < 1e-6
