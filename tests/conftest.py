import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists('/dev/kfd')


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def built():
    """Make sure both shared libraries exist (builds them if a toolchain is present)."""
    import subprocess
    subprocess.run(['make', '-C', os.path.join(REPO, 'oracle'), '-s'], check=True)
    lib = os.path.join(REPO, 'cuburn_amd', '_lib', 'libflame_hip.so')
    if not os.path.isfile(lib):
        subprocess.run(['make', '-C', os.path.join(REPO, 'cuburn_amd', 'csrc'), '-s', '-j4'], check=True)
    return True
