"""The product library loads, exports every symbol include/kindel_hip.h declares, and fails loudly
(no CPU fallback) when there is no GPU.  No compute calls here."""
import os
import re

import pytest

from kindel_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kindel_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kd_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(N.ABI_SYMBOLS)


def test_library_exports_every_symbol():
    lib = N.default_library()
    for sym in header_symbols():
        assert hasattr(lib.dll, sym), sym
    assert lib.dll.kd_abi_version() == 2


def test_emulator_library_exports_the_same_abi(emu_lib):
    for sym in header_symbols():
        assert hasattr(emu_lib.dll, sym), sym


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception) as ei:
        N.Engine([1000])
    assert "kd_create failed" in str(ei.value)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(ImportError):
        N.Library(str(tmp_path / "nope.so"))
