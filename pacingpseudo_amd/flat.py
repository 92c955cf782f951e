"""Flat fp32 parameter / gradient slabs.

All trainable parameters of the model live back to back in ONE device allocation (``params``) and their
gradients in a second one of identical layout (``grads``).  ``nn.Parameter.data`` / ``.grad`` are views into the
slabs, so ``state_dict()`` / ``load_state_dict()`` / ``torch.save`` keep working with the reference's key layout
(train_chaos.py:405-413), while the fused Adam kernel and the RCCL all-reduce see two contiguous arrays.

Segments ("backbone", "aux_path") start on 16-byte boundaries: a segment whose parameters received no gradient in
a step (the auxiliary path in ``--session=Control``) is skipped by the optimiser exactly as ``torch.optim.Adam``
skips parameters whose ``.grad`` is None (train_chaos.py:219).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch


class FlatSlab:
    ALIGN = 4     # floats

    def __init__(self, segments: Sequence[Tuple[str, List[torch.nn.Parameter]]]):
        plist = [p for _, ps in segments for p in ps]
        if not plist:
            raise ValueError('no parameters')
        dev = plist[0].device
        for p in plist:
            if p.device != dev or p.dtype != torch.float32:
                raise TypeError('FlatSlab needs float32 parameters on one device')
        off = 0
        self.segments: Dict[str, Tuple[int, int]] = {}
        self.offsets: Dict[torch.nn.Parameter, int] = {}
        self.seg_params: Dict[str, List[torch.nn.Parameter]] = {}
        for name, ps in segments:
            off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            start = off
            for p in ps:
                self.offsets[p] = off
                off += p.numel()
            end = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.segments[name] = (start, end)
            self.seg_params[name] = list(ps)
            off = end
        self.numel = off
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad_views: Dict[torch.nn.Parameter, torch.Tensor] = {}
        with torch.no_grad():
            for p, o in self.offsets.items():
                view = self.params[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._pp_flat = self
                self.grad_views[p] = self.grads[o:o + p.numel()].view(p.shape)
        self._first = plist[0]
        self.version = 0          # bumped by the optimiser after every in-place update
        # overflow guard of the 16-bit storage mode: [0] = the gradients written by the last backward are not finite
        # (set by pp_scale_guard, reset by the next backward), [1] = optimizer updates skipped for that reason so far
        self.guard = torch.zeros(2, device=dev, dtype=torch.int32)
        self.guard_on = False     # set by the model once a backward ran with a loss scale (fp32 steps never consult the flag)

    def owns(self, p: torch.nn.Parameter) -> bool:
        o = self.offsets.get(p)
        return o is not None and p.data_ptr() == self.params.data_ptr() + 4 * o

    def segment(self, name: str, which: str = 'params') -> torch.Tensor:
        a, b = self.segments[name]
        return getattr(self, which)[a:b]

    def publish_grads(self, active: Sequence[str]) -> None:
        """Make ``p.grad`` the slab view for every parameter of the active segments."""
        for name in active:
            for p in self.seg_params[name]:
                gv = self.grad_views[p]
                if p.grad is not gv:
                    p.grad = gv
