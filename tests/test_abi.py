"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/partls.h declares;
argument validation and the 'no device => loud failure' contract (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "partls.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(partls_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(partls):
    syms = _header_symbols()
    assert len(syms) >= 15
    lib = C.CDLL(partls.library_path())
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/partls.h but not exported"
    bound = {name for name, _, _ in partls.lowlevel.SYMBOLS}
    assert set(syms) == bound, "ctypes binding table and header disagree"


def test_version_and_error_string(partls):
    lib = partls.lowlevel.lib()
    assert lib.partls_version() >= 100
    assert isinstance(lib.partls_last_error(), bytes)


def test_synth_truth_matches_oracle(partls, oracle):
    """host-side part of the generator (partition + true weights) is identical in library and oracle"""
    P, ws = partls.synth_truth(20260003, 37, 5)
    _, _, Po, wo = oracle.synth(20260003, 8, 37, 5, want_X=False)
    assert np.array_equal(P, Po) and np.array_equal(ws, wo)


def test_interface_mirrors_reference_exports(partls):
    # PartitionedLS.jl:3 — export fit, predict, PartLS, PartLSFitResult, Opt, Alt, BnB, regularizeProblem, homogeneousCoords
    for name in ["fit", "predict", "PartLSFitResult", "Opt", "Alt", "BnB", "regularizeProblem", "homogeneousCoords"]:
        assert hasattr(partls, name)
    X = np.arange(6.0).reshape(3, 2)
    P = np.array([[1], [1]])
    Xo, Po = partls.homogeneousCoords(X, P)
    assert Xo.shape == (3, 3) and np.all(Xo[:, -1] == 1) and Po.tolist() == [[1, 0], [1, 0], [0, 1]]
    Xn, yn = partls.regularizeProblem(Xo, np.zeros(3), Po, 4.0)
    assert Xn.shape == (5, 3) and yn.shape == (5,) and Xn[3].tolist() == [2.0, 2.0, 0.0] and Xn[4].tolist() == [0.0, 0.0, 2.0]
    assert partls.regularizeProblem(Xo, np.zeros(3), Po, 0.0)[0] is Xo


def test_argument_errors_are_raised_before_any_device_work(partls):
    X = np.zeros((4, 3)); y = np.zeros(4); P = np.array([[1, 0], [1, 0], [0, 1]])
    with pytest.raises(TypeError):
        partls.fit(partls.Opt, X.astype(int), y, P)
    with pytest.raises(ValueError):
        partls.fit(partls.Opt, X, np.zeros(5), P)
    with pytest.raises(TypeError):
        partls.fit(partls.Opt, X, y, P.astype(float))
    with pytest.raises(TypeError):
        partls.fit(object, X, y, P)


def test_no_gpu_means_loud_failure_not_fallback(partls):
    """Without a HIP device every compute entry must fail with PARTLS_ERR_NO_DEVICE — never silently compute on the CPU."""
    if partls.lowlevel.lib().partls_device_count() > 0:
        pytest.skip("a GPU is present")
    X = np.array([[1., 2, 3], [3, 3, 4], [8, 1, 3], [5, 3, 1]]); y = np.array([1., 1, 2, 3]); P = np.array([[1, 0], [1, 0], [0, 1]])
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.Opt, X, y, P)
    assert ei.value.status == partls.lowlevel.ERR_NO_DEVICE


def test_multi_device_entry_without_gpu(partls):
    """partls_multi_create is part of the same contract: no device => PARTLS_ERR_NO_DEVICE, and RCCL is not loaded for it"""
    lib = partls.lowlevel.lib()
    if lib.partls_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.partls_multi_create(None, 0, C.byref(h)) == partls.lowlevel.ERR_NO_DEVICE and not h
    assert lib.partls_multi_size(None) == 0 and lib.partls_multi_uses_rccl(None) == 0
    import sys
    if "torch" not in sys.modules:                               # PyTorch brings its own copy of RCCL into the process
        assert "librccl" not in open("/proc/self/maps").read()
