"""CPU tests of the Julia side of the boundary: no Julia exists in this image, so the patch is checked
statically — every `ccall` tuple against the prototypes of include/partls.h (tools/check_julia_binding.py), and the ctypes table
the GPU tests call through against the same prototypes, so that Julia, Python and C describe one ABI."""
import ctypes as C
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_julia_binding as CJ  # noqa: E402


def test_every_ccall_of_the_patch_matches_the_header():
    done = CJ.check()
    syms = {s for s, _ in done}
    # the five entry points the patch replaces method bodies with (PartitionedLS.jl:324-328 reaches them through fit/predict)
    for need in ("partls_fit_opt", "partls_fit_opt_multi", "partls_fit_alt", "partls_fit_bnb", "partls_predict",
                 "partls_create", "partls_destroy", "partls_device_count", "partls_last_error", "partls_opt_finish"):
        assert need in syms, f"INTEGRATION.md has no ccall of {need}"


@pytest.mark.parametrize("old,new,what", [
    ("Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Int64}, Int64, Int64, Float64, UInt32,",
     "Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Int64}, Int64, Int64, Float64, Int64,", "argument 11"),
    ("(Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}),",
     "(Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int64}),", "argument types"),
    ("(:partls_predict, _PARTLS_LIB), Cint,", "(:partls_predict, _PARTLS_LIB), Cvoid,", "return type"),
    ("(:partls_fit_bnb, _PARTLS_LIB)", "(:partls_fit_bnb2, _PARTLS_LIB)", "no such symbol"),
])
def test_the_checker_catches_a_wrong_tuple(tmp_path, old, new, what):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert old in text
    p = tmp_path / "INTEGRATION.md"
    p.write_text(text.replace(old, new, 1))
    with pytest.raises(AssertionError) as ei:
        CJ.check(integration=str(p))
    assert what in str(ei.value)


def test_patch_serialises_the_shared_context():
    """a context is not thread-safe (partls.h): the patch must take a lock around every use of the global one"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "ReentrantLock" in text
    blocks = [b for b in text.split("```julia")[1:]]
    for b in blocks:
        code = b.split("```")[0]
        if "_partls_ctx[]" in code and "function fit" in code or "function predict" in code:
            assert "lock(_partls_lock)" in code, "a fit/predict body uses _partls_ctx[] outside lock(_partls_lock)"


_C_FOR_CTYPES = {
    "partls_ctx*": {C.c_void_p}, "partls_ctx**": {C.POINTER(C.c_void_p)},
    "partls_multi*": {C.c_void_p}, "partls_multi**": {C.POINTER(C.c_void_p)},
    "partls_frontier*": {C.c_void_p}, "partls_frontier**": {C.POINTER(C.c_void_p)},
    "double*": {C.POINTER(C.c_double), C.c_void_p}, "int64_t*": {C.POINTER(C.c_int64), C.c_void_p},
    "uint64_t*": {C.POINTER(C.c_uint64)}, "int32_t*": {C.POINTER(C.c_int32)}, "int*": {C.POINTER(C.c_int)},
    "double": {C.c_double}, "int64_t": {C.c_int64}, "uint64_t": {C.c_uint64}, "uint32_t": {C.c_uint32}, "int": {C.c_int},
    "partls_timer": {C.c_int}, "partls_status": {C.c_int}, "char*": {C.c_char_p}, "void": {None},
}


def test_ctypes_table_matches_the_header(partls):
    protos = CJ.parse_header()
    table = {name: (res, args) for name, res, args in partls.lowlevel.SYMBOLS}
    assert set(protos) == set(table)
    for name, (cret, cparams) in protos.items():
        res, args = table[name]
        assert res in _C_FOR_CTYPES[cret], f"{name}: restype {res} vs C '{cret}'"
        assert len(args) == len(cparams), f"{name}: {len(args)} argtypes, {len(cparams)} parameters"
        for k, (a, ct) in enumerate(zip(args, cparams)):
            assert a in _C_FOR_CTYPES[ct], f"{name}: argument {k + 1} is {a}, the prototype says '{ct}'"
