import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def _gpu_run(config):
    return "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")


def pytest_runtest_logstart(nodeid, location):
    """On a GPU run the log names the test that is running BEFORE it runs (flushed): a HIP abort takes the
    whole interpreter down, and the last line of the log must say where.  (faulthandler is switched off for GPU runs only --
    pytest_collection_modifyitems -- so that CPython's 5 KB extension-module dump does not bury that line; CPU runs keep it: a
    native crash in the emulator or the decoder leaves its Python traceback.)"""
    if _GPU_RUN:
        sys.__stdout__.write("\n[gpu-test] %s " % nodeid)
        sys.__stdout__.flush()


_GPU_RUN = False


def pytest_collection_modifyitems(config, items):
    """GPU runs: tests that start other processes, or exchange between ranks go last, so a fault
    in the riskiest part of the runtime cannot erase the parity results in front of it."""
    global _GPU_RUN
    _GPU_RUN = _gpu_run(config)
    if not _GPU_RUN:
        return
    import faulthandler
    faulthandler.disable()
    risky = ("two_ranks", "executable", "subprocess", "graph", "multirank", "test_packaging", "integration_stub", "test_cli")
    items.sort(key=lambda it: any(w in it.nodeid.lower() for w in risky))


@pytest.fixture(scope="session")
def emu_lib():
    """TEST INFRASTRUCTURE: the C-ABI compiled against the CPU kernel emulator (tests/emu)."""
    import __graft_entry__ as g
    from kindel_amd import _native as N
    # KD_EMU_LIB: another build of the same emulator library (scripts/exp/asan_emu.sh: AddressSanitizer + garbage-filled allocations)
    return N.Library(os.environ.get("KD_EMU_LIB") or g.build_emu())


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
