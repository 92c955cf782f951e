#!/usr/bin/env python
"""Round-4 golden vectors, produced by RUNNING THE REFERENCE ITSELF on CPU in the build container:

    python tests/golden/make_golden_r4.py

  aug_ref_elastic.npz   whole samples of the reference's two-stream dataset class (datasets/chaos/chaos_dataset.py:
                        CHAOSTwoStream with chaos_aug_configs.TransformsColor, 64x64 crop) in which ElasticTransform
                        (datasets/augmentations.py:232-277: scipy.ndimage.map_coordinates, order 3 / 0, mode 'nearest')
                        FIRED and neither Scaling nor RandomRotation did -- aug_ref.npz of round 3 holds a single such
                        sample.  The slices are the four files of aug_ref.npz; per sample: seed, file index, the logged
                        random draws (scalars and the two rand(h, w) fields) and the seven output tensors.

Same harness as make_golden_r3.py (placeholder modules for the absent cv2 / skimage, imported from there).  Only DATA is written.
"""
import copy
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_r3 as M  # noqa: E402  (sets up sys.modules and imports the reference)


def main(want=8):
    src = np.load(os.path.join(HERE, 'aug_ref.npz'))
    K = 5
    tmp = tempfile.mkdtemp(prefix='pp_aug4_')
    files, out = [], {}
    for i in range(4):
        f = os.path.join(tmp, f's{i}.npz')
        np.savez(f, uid=f's{i}', img=src[f'files/{i}/img'], lab=src[f'files/{i}/lab'], scb=src[f'files/{i}/scb'])
        files.append(f)
    tr = M.RC.TransformsColor(strength=M.RC.STRENGTH)
    base = copy.deepcopy(tr.base_transforms)
    assert isinstance(base[-1], M.RA.RandomCrop)
    base[-1].crop_size = (64, 64)
    ds = M.CHAOSTwoStream(files, K, base_transforms=base, strong_transforms=tr.strong_transforms, do_strong=True)
    kept, seeds = 0, []
    for seed in range(100, 1200):
        item = seed % len(files)
        np.random.seed(seed)
        try:
            with M.DrawLog() as dl:
                r = ds[item]
        except AttributeError:               # Scaling / RandomRotation reached their placeholder library
            continue
        fields = [a for n, a in dl.log if n == 'rand' and np.ndim(a) == 2]
        if len(fields) != 2:                 # ElasticTransform did not fire
            continue
        p = f'sample/{kept}'
        out[p + '/seed'], out[p + '/item'] = np.asarray(seed), np.asarray(item)
        dl.store(out, p)
        for k in ('image', 'label', 'scribble', 'valid_mask', 'image_strong', 'label_strong', 'scribble_strong'):
            out[p + '/out/' + k] = r[k].numpy()
        seeds.append(seed)
        kept += 1
        if kept == want:
            break
    out['sample/count'] = np.asarray(kept)
    print(f'elastic samples kept: {kept} (seeds {seeds})')
    M.dump(os.path.join(HERE, 'aug_ref_elastic.npz'), out)


if __name__ == '__main__':
    main()
