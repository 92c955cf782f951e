"""Helpers shared by the golden-vector tests: load a captured sequence and replay it."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from oracle import pacing_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TINY = dict(init_ch=4, max_ch=32, hid_ch=8, feat_ch=[32, 32])
FULL = dict(do_loss_ent=True, do_decoder_consistency=True, do_aux_path=True, do_memory=True)

# name -> (args overrides, epochs)   (mirrors tests/golden/make_golden.py:main)
CASES = {
    'full_seq': (dict(**FULL), [0, 0, 1]),
    'control_seq': (dict(), [0]),
    'variant_l1': (dict(**FULL, loss_cr_variants='l1_loss'), [100]),
    'variant_l2': (dict(**FULL, loss_cr_variants='l2_loss', detach_weak_cr=True), [100]),
    'variant_kl': (dict(**FULL, loss_cr_variants='kl_loss', ensemble_mode='mean'), [37, 37]),
    'stride16': (dict(**FULL, output_stride=16), [0]),
    'stride32': (dict(do_loss_ent=True, do_decoder_consistency=True, output_stride=32), [0]),
}


def case_args(name) -> SimpleNamespace:
    over, _ = CASES[name]
    return O.default_args(**TINY, **over)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def to_state(np_sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in np_sd.items()}


def batch_of(d, i):
    return {k: torch.from_numpy(np.array(v)) for k, v in sub(d, f'step{i}/in/').items()}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))


def is_bias_before_bn(key: str) -> bool:
    """Conv biases that feed a BatchNorm: their gradient is exactly 0 in train-mode BN (noise only)."""
    return key.endswith('.conv.bias') or key == 'aux_path.layer_bottleneck.1.bias'
