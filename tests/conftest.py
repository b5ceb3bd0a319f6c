import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device: the gpu-marked tests are skipped, not failed (the driver
    selects them with -m gpu on the GPU box, where nothing is skipped)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_libs_are_built():
    """Build librsx_hip.so / the oracle when missing or older than their sources (hipcc
    cross-compiles without a GPU; a prebuilt, up-to-date library is used as is)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O
