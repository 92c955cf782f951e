"""Auxiliary path + class-prototype memory bank (drop-in for the reference's models/aux_path_memory.py).

Parameter holder only: ``layer_bottleneck`` / ``fc_cls`` / ``memory_bank`` keep the reference's names and
shapes (aux_path_memory.py:21-43) so checkpoints interchange; the arithmetic (3x3 conv + BN + LeakyReLU,
Dropout2d masks, 1x1 classifier, x8 bilinear up-sampling fused with partial CE, memory update of batch sample 0, bank
classification) is executed by ``pacingpseudo_amd.engine.StepEngine`` in HIP kernels.
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
        raise RuntimeError('AuxPath holds parameters only; it runs inside ConsistencyRegulr.forward')
