"""Input pipeline on the GPU through the C ABI against oracle/augment_oracle.py: bit-exact where the
reference's transform is an index permutation (Mirroring, RandomCrop, padding, valid mask, one-hot), float32-rounding
close for the statistics-driven intensity maps, and against the oracle's restatement of the single bicubic resampling
for Scaling / rotation / elastic warps."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO

pytestmark = pytest.mark.gpu


def _slices(rng, B, Hp, Wp, K, sizes=None):
    img = (rng.normal(size=(B, Hp, Wp)) * 30 + 80).astype(np.float32)
    yy, xx = np.mgrid[0:Hp, 0:Wp]
    lab = np.stack([((yy // (9 + n) + xx // (7 + n)) % K) for n in range(B)]).astype(np.int32)
    scb = np.where(rng.uniform(size=(B, Hp, Wp)) < 0.05, lab, K).astype(np.int32)
    img += lab * 15
    if sizes is not None:
        for n, (h, w) in enumerate(sizes):
            for a in (img, lab, scb):
                a[n, h:, :] = 0
                a[n, :, w:] = 0
    return img, lab, scb


def _run(cfg, img, lab, scb, sizes, seed, tweak=None):
    from pacingpseudo_amd.augment import DeviceAugmenter
    aug = DeviceAugmenter(cfg, 'cuda', seed)
    samples = aug.draw(sizes)
    if tweak:
        for p in samples:
            tweak(p)
    out = aug.apply(torch.from_numpy(img), torch.from_numpy(lab), torch.from_numpy(scb), samples)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, aug.last_params['packed'], samples


def test_index_permuting_transforms_are_bit_exact_vs_reference_arithmetic():
    """Mirroring + RandomCrop / padding on ragged slices: the single resampling must be np.flip + the crop copy."""
    from pacingpseudo_amd.augment import AugConfig
    rng = np.random.RandomState(0)
    K, B = 5, 6
    sizes = [(256, 256), (200, 300), (300, 210), (320, 330), (180, 150), (256, 300)]
    img, lab, scb = _slices(rng, B, 320, 330, K, sizes)
    cfg = AugConfig(num_classes=K, p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, do_strong=False)
    out, pk, samples = _run(cfg, img, lab, scb, sizes, 11)
    assert any(p['flip0'] for p in samples) and any(p['flip1'] for p in samples)
    for n, (h, w) in enumerate(sizes):
        p = samples[n]
        ri = AO.mean_std_norm(img[n, :h, :w].astype(np.float64))
        rl, rs = lab[n, :h, :w], scb[n, :h, :w]
        if p['flip0']:
            ri, rl, rs = AO.mirroring([ri, rl, rs], 0)
        if p['flip1']:
            ri, rl, rs = AO.mirroring([ri, rl, rs], 1)
        # (the reference normalises once more before the crop; that commutes with the index permutation)
        ri = AO.mean_std_norm(ri)
        ci, cl, cs, cv = AO.random_crop(ri, rl, rs, cfg.crop_size, p['image_top'], p['image_left'], p['canvas_top'],
                                        p['canvas_left'], 0, K)
        np.testing.assert_array_equal(out['valid_mask'][n, 0], cv)
        np.testing.assert_array_equal(out['label'][n], AO.to_one_hot(cl, K))
        np.testing.assert_array_equal(out['scribble'][n], AO.to_one_hot(cs, K + 1))
        if h <= 256 and w <= 256:                       # whole slice visible: statistics over the same pixels
            np.testing.assert_allclose(out['image'][n, 0], ci, atol=3e-5)
        else:                                           # cropped: same pixels up to the second normalisation's window
            m = cv > 0
            z = out['image'][n, 0][m]
            assert abs(z.mean()) < 1e-4 and abs(z.std() - 1) < 1e-3
            a, b = np.polyfit(ci[m], z, 1)
            np.testing.assert_allclose(a * ci[m] + b, z, atol=3e-4)
    # ignored padding: label 'K' has no one-hot plane, scribble K is the ignored plane
    n = 4
    assert out['label'][n][:, out['valid_mask'][n, 0] == 0].sum() == 0
    assert (out['scribble'][n][K][out['valid_mask'][n, 0] == 0] == 1).all()


def test_statistics_and_scalar_maps_match_oracle():
    from pacingpseudo_amd._lib import lib
    rng = np.random.RandomState(1)
    B, H, W = 5, 97, 131
    x = (rng.normal(size=(B, H, W)) * 20 + 50).astype(np.float32)
    rect = np.array([[0, 0, H, W], [3, 5, 50, 60], [10, 0, 1, W], [0, 7, H, 1], [40, 40, 57, 91]], np.int32)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(rect).cuda()
    st = torch.empty(B, 4, dtype=torch.float64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    lib.pp_aug_stats(xd.data_ptr(), B, H, W, rd.data_ptr(), st.data_ptr(), s)
    want = np.stack([AO.stats(x[n], rect[n]) for n in range(B)])
    np.testing.assert_allclose(st.cpu().numpy(), want, rtol=1e-12, atol=1e-9)
    coef = torch.empty(B, 4, device='cuda')
    par = torch.tensor([1.3, AO.SKIP, 0.4, 1.7, 0.9], device='cuda')
    for mode in (0, 1, 2, 4):
        lib.pp_aug_coef(st.data_ptr(), None, par.data_ptr() if mode else None, mode, B, coef.data_ptr(), s)
        wc = np.stack([AO.coef(mode, want[n], None, par[n].item() if mode else None) for n in range(B)])
        np.testing.assert_allclose(coef.cpu().numpy(), wc, rtol=1e-6)
        y = xd.clone()
        if mode == 2:
            lib.pp_aug_gamma(y.data_ptr(), B, H, W, coef.data_ptr(), rd.data_ptr(), s)
        else:
            lib.pp_aug_scalar_map(y.data_ptr(), B, H, W, coef.data_ptr(), rd.data_ptr(), s)
        for n in range(B):
            t, l, h, w = rect[n]
            wy = x[n].copy()
            if mode == 2:
                wy[t:t + h, l:l + w] = AO.gamma_map(x[n, t:t + h, l:l + w], wc[n])
            else:
                wy = AO.scalar_map(x[n], wc[n], rect[n])
            np.testing.assert_allclose(y[n].cpu().numpy(), wy, rtol=2e-5, atol=1e-5)


def test_noise_and_elastic_fields_match_oracle():
    from pacingpseudo_amd._lib import lib
    B, H, W = 3, 64, 80
    s = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(B, H, W, device='cuda')
    sig = torch.tensor([0.1, 0.0, 2.0], device='cuda')
    lib.pp_aug_add_noise(x.data_ptr(), B, H, W, sig.data_ptr(), None, 777, s)
    z = AO.normal_field(B, H * W, 777).reshape(B, H, W) * sig.cpu().numpy()[:, None, None]
    np.testing.assert_allclose(x.cpu().numpy(), z, atol=2e-5 * 2.0 * 5)
    assert float(x[1].abs().max()) == 0
    sa = np.array([[9.5, 150.0], [0.0, 0.0], [12.7, 40.0]], np.float32)
    disp, scr = torch.empty(B, 2, H, W, device='cuda'), torch.empty(B, 2, H, W, device='cuda')
    lib.pp_aug_elastic_field(disp.data_ptr(), scr.data_ptr(), B, H, W, torch.from_numpy(sa).cuda().data_ptr(), 4242, s)
    want = AO.device_elastic_field(B, H, W, sa, 4242)
    np.testing.assert_allclose(disp.cpu().numpy(), want, atol=2e-4)
    assert float(disp[1].abs().max()) == 0 and float(disp[0].abs().max()) > 0.5


@pytest.mark.parametrize('seed', [0, 1])
def test_full_two_stream_batch_matches_oracle_scaling_rotation_unpinned(seed):
    """Everything switched on with high probabilities, ragged slices, against the oracle pipeline."""
    from pacingpseudo_amd.augment import AugConfig
    rng = np.random.RandomState(seed)
    K, B = 5, 8
    sizes = [(256, 256), (240, 272), (288, 250), (200, 200), (256, 256), (310, 300), (256, 256), (230, 256)]
    img, lab, scb = _slices(rng, B, 310, 300, K, sizes)
    cfg = AugConfig(num_classes=K, p_scaling=0.6, p_elastic=0.5, p_rotation=0.6, p_noise=0.5)
    out, pk, samples = _run(cfg, img, lab, scb, sizes, 100 + seed)
    assert any(p['scale'] is not None for p in samples) and any(p['degree'] is not None for p in samples)
    assert any(p['sigma'] > 0 for p in samples) and any(p['noise'] > 0 for p in samples)
    want = AO.pipeline(img, lab, scb, pk, cfg.crop_size, K)
    np.testing.assert_array_equal(out['valid_mask'], want['valid_mask'])
    # nearest-neighbour class maps: a coordinate within float rounding of a pixel boundary may fall either side
    for k in ('label', 'scribble'):
        assert (out[k] != want[k]).mean() < 2e-4, k
    for k in ('image', 'image_strong'):
        d = np.abs(out[k] - want[k])
        assert np.median(d) < 1e-5 and (d > 2e-3).mean() < 2e-4, (k, np.median(d), (d > 2e-3).mean(), d.max())
    for n in range(B):
        m = out['valid_mask'][n, 0] > 0
        z = out['image'][n, 0][m]
        assert abs(z.mean()) < 1e-3 and abs(z.std() - 1) < 1e-2
        assert (out['image'][n, 0][~m] == 0).all()
    # strong stream = Brightness -> Contrast -> Gamma of the weak image, in the reference's float64 arithmetic
    for n, p in enumerate(samples):
        s = out['image'][n, 0].astype(np.float64)
        if p['bright'] > AO.SKIP:
            s = AO.brightness(s, p['bright'])
        if p['contrast'] > AO.SKIP:
            s = AO.contrast(s, p['contrast'])
        if p['gamma'] > AO.SKIP:
            s = AO.gamma_augmentation(s, p['gamma'])
        np.testing.assert_allclose(out['image_strong'][n, 0], s, atol=5e-4)


def test_rotation_is_the_cv2_convention_unpinned():
    """A +90 degree cv2 rotation about (w/2, h/2) sends a marker at (y, x) to (h/2 + w/2 - x... ) -- checked on one pixel:
    warpAffine samples src at M^-1(dst), and for angle > 0 the content turns counter-clockwise (cv2 docs)."""
    from pacingpseudo_amd.augment import AugConfig
    K = 5
    img = np.zeros((1, 256, 256), np.float32)
    lab = np.zeros((1, 256, 256), np.int32)
    lab[0, 100, 200] = 3                                   # marker right of the centre, slightly above it
    cfg = AugConfig(num_classes=K, p_scaling=0, p_elastic=0, p_rotation=1.0, p_noise=0, do_strong=False)

    def tweak(p):
        p.update(degree=90.0, flip0=False, flip1=False)
    out, _, _ = _run(cfg, img + np.random.RandomState(0).normal(size=img.shape).astype(np.float32), lab, lab.copy(), [(256, 256)], 3, tweak)
    ys, xs = np.nonzero(out['label'][0, 3])
    # counter-clockwise by 90 degrees about (128, 128): (dy, dx) = (-28, +72) -> (-72, -28)
    assert len(ys) == 1 and (int(ys[0]), int(xs[0])) == (128 - 72, 128 - 28)


def test_training_step_consumes_an_augmented_batch():
    from pacingpseudo_amd.augment import AugConfig, DeviceAugmenter, collate_raw
    from pacingpseudo_amd.data import SyntheticPhantoms
    from oracle import pacing_oracle as O
    from tests.test_gpu_step import build_model
    args = O.full_flags(epoch=2)
    sd = O.init_state(args, seed=1)
    model = build_model(args, {k: v.numpy() for k, v in sd.items()})
    model.train()
    ds = SyntheticPhantoms(4, args.num_classes, size=96, raw=True, seed=2)
    b = collate_raw([ds[i] for i in range(4)])
    aug = DeviceAugmenter(AugConfig(num_classes=args.num_classes, crop_size=(64, 64)), 'cuda', 5)
    batch = aug(b['img'], b['lab'], b['scb'], b['sizes'])
    assert batch['image'].shape == (4, 1, 64, 64) and batch['scribble'].shape == (4, args.num_classes + 1, 64, 64)
    batch.pop('label'); batch.pop('label_strong')
    out = model(batch, mode='train', step=0)
    loss = out['loss_pce'] + out['loss_cr'] + out['loss_aux_cls']
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(model.flat.grads).all()


def _colour_in_f64(weak, p):
    s = weak.astype(np.float64)
    if p['bright'] > AO.SKIP:
        s = AO.brightness(s, p['bright'])
    if p['contrast'] > AO.SKIP:
        s = AO.contrast(s, p['contrast'])
    if p['gamma'] > AO.SKIP:
        s = AO.gamma_augmentation(s, p['gamma'])
    return s


@pytest.mark.parametrize('recipe', ['TransformsColorBlur', 'TransformsColorMixup',
                                    pytest.param('TransformsColorLow', id='TransformsColorLow-lowres_unpinned')])
def test_strong_view_recipes(recipe):
    """The fourth strong transform of chaos_aug_configs.py:88-186 on top of the colour transforms: GaussianBlur and Mixup
    against the reference's arithmetic (scipy's gaussian_filter; the lam-blend with the centre-cropped, normalised partner),
    SimulationLowRes against the oracle's restatement of the nearest-down / cubic-up resampling."""
    from pacingpseudo_amd.augment import AugConfig
    rng = np.random.RandomState(7)
    K, B = 5, 4
    sizes = [(256, 256), (300, 280), (256, 256), (272, 256)]
    img, lab, scb = _slices(rng, B, 300, 280, K, sizes)
    cfg = AugConfig(num_classes=K, p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, recipe=recipe, p_extra=1.0)
    out, pk, samples = _run(cfg, img, lab, scb, sizes, 21)
    for n, p in enumerate(samples):
        base = _colour_in_f64(out['image'][n, 0], p)
        got = out['image_strong'][n, 0]
        if recipe == 'TransformsColorBlur':
            assert p['blur'] >= 1.0
            np.testing.assert_allclose(got, AO.gaussian_blur(base, p['blur']), atol=5e-4)
        elif recipe == 'TransformsColorMixup':
            assert 0.8 <= p['lam'] <= 1.0 and 0 <= p['partner'] < B
            h2, w2 = sizes[p['partner']]
            partner = AO.center_crop(img[p['partner'], :h2, :w2].astype(np.float64), 256, 256)
            np.testing.assert_allclose(got, AO.mixup(base, partner, p['lam']), atol=5e-4)
        else:
            assert 1.5 <= p['lowres'] <= 2.0
            want = AO.lowres(base.astype(np.float32), p['lowres'], AO.stats(base.astype(np.float32)))
            d = np.abs(got - want)
            assert np.median(d) < 1e-4 and (d > 5e-3).mean() < 1e-3, (np.median(d), (d > 5e-3).mean(), d.max())
            assert np.abs(got - base).mean() > 1e-3                    # it did lose resolution


def test_mixup_partner_from_the_whole_file_list():
    """Mixup's second image comes from the WHOLE file list in the reference (datasets/augmentations.py:66:
    `np.load(np.random.choice(file_ls))`, handed in by chaos_dataset.py:73-74).  The raw loader supplies one candidate per
    sample (data.NpzSlices.mix_partner -> collate_raw -> DeviceAugmenter(mix=...)); the blend must use THAT slice -- of another
    size than any slice of the batch here -- centre-cropped and normalised as augmentations.py:66-71 does."""
    from pacingpseudo_amd.augment import AugConfig, DeviceAugmenter
    rng = np.random.RandomState(17)
    K, B = 5, 3
    sizes = [(256, 256)] * B
    img, lab, scb = _slices(rng, B, 256, 256, K, sizes)
    msizes = [(300, 280), (256, 256), (272, 290)]
    mix = np.zeros((B, 300, 290), np.float32)
    for n, (h, w) in enumerate(msizes):
        mix[n, :h, :w] = rng.normal(size=(h, w)) * 30 + 80
    cfg = AugConfig(num_classes=K, p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, recipe='TransformsColorMixup', p_extra=1.0)
    aug = DeviceAugmenter(cfg, 'cuda', 5)
    samples = aug.draw(sizes)
    out = aug.apply(torch.from_numpy(img), torch.from_numpy(lab), torch.from_numpy(scb), samples,
                    mix=(torch.from_numpy(mix), msizes))
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for n, p in enumerate(samples):
        assert 0.8 <= p['lam'] <= 1.0
        base = _colour_in_f64(out['image'][n, 0], p)
        h2, w2 = msizes[n]
        partner = AO.center_crop(mix[n, :h2, :w2].astype(np.float64), 256, 256)
        np.testing.assert_allclose(out['image_strong'][n, 0], AO.mixup(base, partner, p['lam']), atol=5e-4)
