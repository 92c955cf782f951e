import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module', autouse=True)
def _release_gpu_memory_between_modules():
    """The GPU suite runs in ONE process (the box allows few processes on its card): after a module, give the caching allocator's
    blocks back.  The batch-32 parity cases leave ~32 GB of cached plan buffers behind, and every DataLoader worker a later
    test forks has to duplicate the page tables of whatever the process has mapped -- the training-driver test of round 5 took
    156 s of the suite's 523 s that way (profiles/r06_pytest_gpu_tail.log has the durations before / after)."""
    yield
    import gc
    gc.collect()
    if 'torch' in sys.modules:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
