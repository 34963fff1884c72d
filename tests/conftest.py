import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["split16", "f32"])
def gemm_mode(request):
    """Both ways of forming the hidden-layer products (include/isopoints.h: iso_siren_set_gemm_mode)."""
    from iso_points_amd import _lib
    lib = _lib.load()
    before = lib.iso_siren_get_gemm_mode()
    _lib.call("iso_siren_set_gemm_mode", 1 if request.param == "split16" else 0)
    yield request.param
    _lib.call("iso_siren_set_gemm_mode", before)
