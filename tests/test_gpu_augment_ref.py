"""The device input pipeline against PIXELS OF THE REFERENCE ITSELF (tests/golden/aug_ref.npz, captured by
tests/golden/make_golden_r3.py from datasets/augmentations.py and datasets/chaos/chaos_dataset.py:CHAOSTwoStream).

The reference's random draws are replayed through `draw_sample`; its ElasticTransform / GaussianNoise fields are handed to
`DeviceAugmenter.apply(fields=...)`.  What must hold:
  * valid mask, class maps: bit-exact, with and without ElasticTransform (round 4: displaced coordinates in double);
  * weak image: float32-rounding close; ElasticTransform samples are resampled with scipy's own interpolant -- the prefiltered
    cubic B-spline of map_coordinates(order=3, mode='nearest') -- on the device (round 4; round 3 used the Keys kernel there);
  * strong image: Brightness -> Contrast -> GammaAugmentation of the weak image, to 5e-4.
Scaling, RandomRotation and SimulationLowRes are absent from the fixtures (skimage / cv2 not installed): UNPINNED, see
tests/test_gpu_augment.py::test_full_two_stream_batch_matches_oracle_scaling_rotation_unpinned for the self-consistency check of those."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO
from tests import _golden as G

pytestmark = pytest.mark.gpu


class ReplayRNG:
    """numpy.random.RandomState stand-in that hands out the scalar draws the reference made, in order.  The two extra
    `randint(2**31 - 1)` calls of draw_sample (device Philox seeds, where the reference draws whole fields) consume nothing."""

    def __init__(self, draws):
        self.d = [float(x) for x in draws]

    def uniform(self, *a):
        return self.d.pop(0)

    def randint(self, n):
        return 0 if n == 2 ** 31 - 1 else int(self.d.pop(0))


@pytest.fixture(scope='module')
def aug():
    return G.load('aug_ref')


@pytest.mark.parametrize('which', ['aug_ref', 'aug_ref_elastic'])
def test_whole_reference_samples_through_the_device_pipeline(aug, which):
    """aug_ref: the twelve samples of round 3 (one with ElasticTransform); aug_ref_elastic (round 4,
    tests/golden/make_golden_r4.py): eight more in which ElasticTransform fired, on the same four slices."""
    from pacingpseudo_amd.augment import AugConfig, DeviceAugmenter, collate_raw, compose_map, draw_sample
    K, crop = 5, (64, 64)
    files = aug
    if which != 'aug_ref':
        aug = dict(G.load(which))
        aug.update({k: v for k, v in files.items() if k.startswith('files/')})
    n = int(aug['sample/count'])
    cfg = AugConfig(num_classes=K, crop_size=crop)
    items, samples, refs = [], [], []
    disp = np.zeros((n, 2) + crop, np.float64)                # the reference's fields are float64 (augmentations.py:264-265)
    noise = np.zeros((n,) + crop, np.float32)
    kinds = []
    for i in range(n):
        p = f'sample/{i}'
        item = int(aug[p + '/item'])
        img, lab, scb = aug[f'files/{item}/img'], aug[f'files/{item}/lab'], aug[f'files/{item}/scb']
        items.append(dict(img=img, lab=lab.astype(np.int32), scb=scb.astype(np.int32)))
        rng = ReplayRNG(aug[p + '/draws'])
        s = draw_sample(rng, img.shape[0], img.shape[1], cfg)
        assert not rng.d, 'every draw of the reference must have been consumed'
        assert s['scale'] is None and s['degree'] is None
        arrs = [aug[f'{p}/draw_arr{j}'] for j in range(4) if f'{p}/draw_arr{j}' in aug]
        m = compose_map(s)
        yo, xo = np.mgrid[0:crop[0], 0:crop[1]]
        inside = (yo >= s['canvas_top']) & (yo < s['canvas_top'] + s['patch_h']) & (xo >= s['canvas_left']) & (xo < s['canvas_left'] + s['patch_w'])
        ys = np.clip(np.rint(m[0] * yo + m[1] * xo + m[2]).astype(int), 0, img.shape[0] - 1)
        xs = np.clip(np.rint(m[3] * yo + m[4] * xo + m[5]).astype(int), 0, img.shape[1] - 1)
        if s['sigma'] > 0:                       # the reference's displacement lives on the SOURCE grid (before the flips)
            dx = AO.elastic_field(arrs.pop(0) * 2 - 1, s['sigma'], s['alpha'])
            dy = AO.elastic_field(arrs.pop(0) * 2 - 1, s['sigma'], s['alpha'])
            disp[i, 0], disp[i, 1] = np.where(inside, dy[ys, xs], 0), np.where(inside, dx[ys, xs], 0)
        if s['noise'] > 0:                       # the noise field lives on the grid after the flips, before the crop
            f = arrs.pop(0)
            yy = np.clip(yo - s['canvas_top'] + s['image_top'], 0, f.shape[0] - 1)
            xx = np.clip(xo - s['canvas_left'] + s['image_left'], 0, f.shape[1] - 1)
            noise[i] = np.where(inside, f[yy, xx], 0)
        kinds.append((s['sigma'] > 0, s['noise'] > 0, img.shape[0] > crop[0] or img.shape[1] > crop[1]))
        samples.append(s)
        refs.append(G.sub(aug, p + '/out/'))
    b = collate_raw(items)
    dev = DeviceAugmenter(cfg, 'cuda', 0)
    out = dev.apply(b['img'], b['lab'], b['scb'], samples, fields=dict(disp=torch.from_numpy(disp), noise=torch.from_numpy(noise)))
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    if which == 'aug_ref':
        assert any(k[0] for k in kinds) and any(k[1] for k in kinds) and any(not k[0] and not k[1] for k in kinds)
    else:
        assert all(k[0] for k in kinds) and any(k[2] for k in kinds) and any(not k[2] for k in kinds)
    worst_plain, report = 0.0, []
    for i, (elastic, noisy, cropped) in enumerate(kinds):
        r = refs[i]
        np.testing.assert_array_equal(out['valid_mask'][i], r['valid_mask'], err_msg=f'sample {i}')
        for k in ('label', 'scribble'):
            # with ElasticTransform too (round 4): the displaced coordinates are formed and rounded in double, as scipy does
            np.testing.assert_array_equal(out[k][i], r[k], err_msg=f'sample {i} {k} (elastic: {elastic})')
            np.testing.assert_array_equal(out[k + '_strong'][i], out[k][i])
        got, want = out['image'][i, 0], r['image'][0]
        mask = r['valid_mask'][0] > 0
        if cropped:
            # the second MeanStdNorm: the reference normalises the whole pre-crop slice, the device the surviving window
            a, c = np.polyfit(want[mask], got[mask], 1)
            want = np.where(mask, a * want + c, 0)
            assert 0.8 < a < 1.25
        d = np.abs(got - want)
        if elastic:
            # scipy's cubic B-spline on the device (pp_aug_spline_prefilter + pp_aug_warp_spline): the reference's pixels, up to
            # the float32 arithmetic of the normalisations around it
            report.append((i, float(np.median(d)), float(np.quantile(d, 0.99)), float(d.max())))
            assert d.max() < (3e-4 if cropped else 5e-6), (i, cropped, d.max())      # measured 3.7e-7 .. 6.1e-7 (r04)
        else:
            worst_plain = max(worst_plain, float(d.max()))
            assert d.max() < (3e-4 if cropped else 5e-5), (i, noisy, cropped, d.max())
        # strong view from the device's own weak image, in the reference's arithmetic
        s = samples[i]
        st = got.astype(np.float32)
        if s['bright'] > AO.SKIP:
            st = AO.brightness(st, float(s['bright']))
        if s['contrast'] > AO.SKIP:
            st = AO.contrast(st, float(s['contrast']))
        if s['gamma'] > AO.SKIP:
            st = AO.gamma_augmentation(st, float(s['gamma']))
        np.testing.assert_allclose(out['image_strong'][i, 0], st, atol=5e-4, err_msg=f'sample {i} strong')
        if not elastic and not cropped:
            np.testing.assert_allclose(out['image_strong'][i, 0], r['image_strong'][0], atol=5e-4, err_msg=f'sample {i} strong vs ref')
    print(f'weak image vs reference pixels, no ElasticTransform: max |d| {worst_plain:.2e}')
    for i, med, p99, mx in report:
        print(f'sample {i} with ElasticTransform (cubic B-spline, as scipy): median |d| {med:.2e}, p99 {p99:.2e}, max {mx:.2e}')


@pytest.mark.parametrize('s', [0, 1])
def test_rotation90_and_cutout_against_reference_pixels(aug, s):
    """The two transforms of augmentations.py no recipe uses (Rotation90 :319-335, Cutout :23-49), opt-in on the device."""
    from pacingpseudo_amd.augment import AugConfig, DeviceAugmenter, cutout_rect, draw_sample
    K = 5
    p = f'rot90/{s}'
    img, lab, scb = aug[p + '/in/image'], aug[p + '/in/label'].astype(np.int32), aug[p + '/in/scribble'].astype(np.int32)
    k = int(aug[p + '/draws'][1])
    h, w = img.shape
    oh, ow = (w, h) if k % 2 else (h, w)
    cfg = AugConfig(num_classes=K, crop_size=(oh, ow), p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, p_mirror=0,
                    do_strong=False, p_rot90=1.0)
    sample = draw_sample(np.random.RandomState(0), h, w, cfg)
    sample['rot90'] = k
    sample['nh'], sample['nw'] = oh, ow
    sample.update(image_top=0, image_left=0, canvas_top=0, canvas_left=0, patch_h=oh, patch_w=ow)
    dev = DeviceAugmenter(cfg, 'cuda', 0)
    out = dev.apply(torch.from_numpy(img[None]), torch.from_numpy(lab[None]), torch.from_numpy(scb[None]), [sample])
    ref = G.sub(aug, p + '/out/')
    np.testing.assert_array_equal(out['label'][0].cpu().numpy(), AO.to_one_hot(ref['label'], K))
    np.testing.assert_array_equal(out['scribble'][0].cpu().numpy(), AO.to_one_hot(ref['scribble'], K + 1))
    np.testing.assert_allclose(out['image'][0, 0].cpu().numpy(), AO.mean_std_norm(ref['image'].astype(np.float64)), atol=3e-5)

    p = f'cutout/{s}'
    img = aug[p + '/in/image']
    h, w = img.shape
    y, x = int(aug[p + '/draws'][1]), int(aug[p + '/draws'][2])
    cfg = AugConfig(num_classes=K, crop_size=(h, w), p_scaling=0, p_elastic=0, p_rotation=0, p_noise=0, p_mirror=0, p_color=0,
                    p_cutout=1.0, cutout_length=16)
    sample = draw_sample(np.random.RandomState(0), h, w, cfg)
    sample['cutout'] = cutout_rect(y, x, 16, h, w)
    z = np.zeros((1, h, w), np.int32)
    out = DeviceAugmenter(cfg, 'cuda', 0).apply(torch.from_numpy(img[None]), torch.from_numpy(z), torch.from_numpy(z), [sample])
    weak = out['image'][0, 0].cpu().numpy()
    np.testing.assert_array_equal(out['image_strong'][0, 0].cpu().numpy(), AO.cutout(weak, 16, y, x))
    # and the reference's own output: the same square of its (already normalised) input is zero, the rest untouched
    ref = aug[p + '/out/image']
    np.testing.assert_array_equal(ref == 0, (AO.cutout(np.ones_like(img), 16, y, x) == 0) | (img == 0))
    np.testing.assert_allclose(out['image_strong'][0, 0].cpu().numpy(), AO.mean_std_norm(img.astype(np.float64)) * (ref != 0), atol=3e-5)
