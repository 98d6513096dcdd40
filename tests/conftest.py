import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries (they are git-ignored): build the product and the checker once
    lib = os.path.join(ROOT, "sparse_matrix_amd", "libspam_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def spam():
    """The product, through its C ABI.  Fails loudly when the HIP library or the GPU is missing."""
    import sparse_matrix_amd as s
    from sparse_matrix_amd import _lib

    _lib.ensure_device()
    return s
