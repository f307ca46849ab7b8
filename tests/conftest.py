import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def emu_lib():
    """TEST INFRASTRUCTURE: the C-ABI compiled against the CPU kernel emulator (tests/emu)."""
    import __graft_entry__ as g
    from kindel_amd import _native as N
    return N.Library(g.build_emu())


@pytest.fixture(scope="session")
def hip_lib():
    """The product library on a real GPU."""
    import torch
    from kindel_amd import _native as N
    if not torch.cuda.is_available():
        pytest.skip("needs a real MI355X (the -m gpu suite runs through gpurun)")
    return N.default_library()


@pytest.fixture()
def api_on_emu(emu_lib, monkeypatch):
    """Route kindel_amd's Python API through the emulator library (CPU logic tests only)."""
    from kindel_amd import _native as N
    monkeypatch.setattr(N, "_default", emu_lib)
    return emu_lib
