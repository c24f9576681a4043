"""CPU: the C-ABI library loads without a GPU and exports every symbol include/semabs.h declares (no compute calls);
the ctypes table covers the same set; host-side entry points (coefficient tables) match the oracle."""
import ctypes
import os
import re

import numpy as np

import semabs_amd  # noqa: F401
from conftest import ROOT
from oracle import preprocess as op


def _build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("semabs_build", os.path.join(ROOT, "semantic-abstraction_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "semabs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(semabs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_header_symbols():
    lib = ctypes.CDLL(_build())
    syms = _header_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from semabs_amd import _lib
    assert sorted(set(_lib.SIGNATURES) | {"semabs_last_error"}) == _header_symbols()
    assert _lib.lib().semabs_abi_version() == 8


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    from semabs_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.require_gpu()
    from semabs_amd.clip import ClipWrapper
    ClipWrapper.engine = None
    with pytest.raises(RuntimeError):
        ClipWrapper("ViT-B/32", state_dict={})


def test_resize_coeffs_host_entry_matches_oracle():
    from semabs_amd import _lib
    for ts in (480, 320, 240, 120, 224, 30, 97):
        xmin = np.zeros(224, np.int32); kk = np.zeros((224, 24), np.int32); ks = ctypes.c_int(0)
        _lib.call("semabs_resize_coeffs", ts, 224, xmin.ctypes.data, kk.ctypes.data, 24, ctypes.addressof(ks))
        oxmin, ocnt, okk = op.resample_coeffs(ts, 224)
        assert ks.value == okk.shape[1] and np.array_equal(xmin, oxmin) and np.array_equal(kk[:, : okk.shape[1]], okk)
        assert not kk[:, okk.shape[1]:].any()


def test_rejects_bad_arguments_without_touching_the_gpu():
    from semabs_amd import _lib
    h = _lib.lib()
    assert h.semabs_gemm_f16(None, None, None, None, None, 4, 128, 64, 64, 64, 128, 3, None, None) == -1
    assert b"null operand" in h.semabs_last_error()
    xmin = np.zeros(224, np.int32); kk = np.zeros((224, 24), np.int32); ks = ctypes.c_int(0)
    assert h.semabs_resize_coeffs(5000, 224, xmin.ctypes.data, kk.ctypes.data, 24, ctypes.addressof(ks)) == -1     # too many taps
