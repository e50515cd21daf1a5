"""The C-ABI library loads without a GPU and exports every symbol include/chest.h declares."""
import ctypes
import importlib
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "chest.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chest_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  -- torch's HIP runtime must be mapped before libchest.so (see _native.lib)
    N = importlib.import_module("channel-estimation-in-5g-network_amd._native")
    assert os.path.exists(N.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(N.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in chest.h but not exported"
    assert set(names) == set(N.SIGNATURES), set(names) ^ set(N.SIGNATURES)


def test_version_and_error_channel_work_without_gpu():
    N = importlib.import_module("channel-estimation-in-5g-network_amd._native")
    L = N.lib()
    assert L.chest_abi_version() == N.ABI_VERSION
    assert L.chest_frames_workspace_bytes(None, 10) == 0
    assert L.chest_mmse_full_workspace_bytes(100) == 16 * (128 * 128 + 2 * 64 * 128 + 2 * 64 * 64)  # padded to the 64-wide pivot blocks; two pivot-inverse buffers (look-ahead)
    h = ctypes.c_void_p()
    bad = N.Config(1000, 72, 14, 600, 15000.0, 2, 2, 0)          # S*nsc fine, but U > fft is rejected before any HIP call
    bad.useful_subcarriers = 5000
    assert L.chest_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b"unsupported configuration" in L.chest_last_error()


def test_qam_tables_equal_numpy_construction():
    """chest_qam_constellation is a host table query (no GPU): every bit equals the NumPy expression of src/utils.py:93-101
    as stored by the golden generator."""
    import numpy as np
    import pytest
    from conftest import GOLDEN
    N = importlib.import_module("channel-estimation-in-5g-network_amd._native")
    L = N.lib()
    for M in (4, 16):
        g = np.load(os.path.join(GOLDEN, f"qam{M}.npz"))
        pts = np.zeros(M, dtype=np.complex128)
        gray = np.zeros(M, dtype=np.int32)
        assert L.chest_qam_constellation(M, pts.ctypes.data, gray.ctypes.data) == 0
        assert np.array_equal(pts.view(np.float64), g["constellation"].view(np.float64))
        assert np.array_equal(gray, g["gray_map"])
    assert L.chest_qam_constellation(64, None, None) == -1
    assert b"not implemented" in L.chest_last_error()
    del pytest
