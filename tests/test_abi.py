"""The C-ABI shared library loads and exports every symbol declared in include/codd_hip.h, and the
ctypes table binds every one of them (no compute calls: no GPU here)."""
import os
import re

from codd_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "codd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(codd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = declared()
    assert len(names) >= 25
    lib = _abi.load()
    assert _abi.MISSING == []
    for n in names:
        assert n in _abi.SIGNATURES, f"{n} declared in the header but not bound in codd_amd/_abi.py"
        assert getattr(lib, n) is not None
    for n in _abi.SIGNATURES:
        assert n in names, f"{n} bound but not declared in include/codd_hip.h"
    assert lib.codd_abi_version() == _abi.ABI_VERSION == 12  # CODD_ABI_VERSION of include/codd_hip.h


def test_conv_packed_size_is_consistent():
    lib = _abi.load()
    # [ncog][nchunks][taps][ck][wrow]: Cout=34 mb=4 -> 1 cog of 64 (+16 pad), Cin=32 ck=16 -> 2 chunks
    assert lib.codd_conv2d_packed_size(34, 32, 3, 3, 4, 16) == 1 * 2 * 9 * 16 * 80
    assert lib.codd_conv2d_packed_size(16, 3, 3, 3, 1, 4) == 1 * 1 * 9 * 4 * 16
    assert lib.codd_conv2d_packed_size(16, 3, 3, 3, 1, 3) == -1


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from codd_amd import ops
    with pytest.raises(_abi.CoddHipError):
        ops.PackedConv(torch.zeros(16, 3, 3, 3), torch.zeros(16))


def test_roll_packed_size_and_argument_checks():
    """codd_roll_packed_size / codd_conv_roll argument validation (no launch: every call below is rejected first)."""
    import ctypes as C
    lib = _abi.load()
    assert lib.codd_roll_packed_size(16, 3, 16) == 1 * 9 * 1 * 64 * 4
    assert lib.codd_roll_packed_size(32, 3, 32) == 2 * 9 * 2 * 64 * 4
    assert lib.codd_roll_packed_size(32, 1, 40) == 2 * 3 * 64 * 4
    assert lib.codd_roll_packed_size(24, 3, 24) == -1
    p = _abi.RollParams()
    assert lib.codd_conv_roll(C.byref(p), None) != 0  # C = 0
    p.C, p.mode = 48, 0
    assert lib.codd_conv_roll(C.byref(p), None) == -2  # CODD_EUNSUPPORTED
    p.C = 16
    assert lib.codd_conv_roll(C.byref(p), None) == -1  # CODD_EINVAL: null pointers
