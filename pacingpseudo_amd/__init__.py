"""pacingpseudo_amd: the PacingPseudo training step (zefanyang/pacingpseudo) rebuilt for AMD MI355X (gfx950).

Python on PyTorch-ROCm is the host language (as in the reference); all arithmetic of the hot path runs in the
hand-written HIP kernels of ``libpacingpseudo_hip.so`` (C ABI: include/pacingpseudo_hip.h)."""
__version__ = '0.1.0'

from ._lib import HipLibraryError, lib  # noqa: F401
