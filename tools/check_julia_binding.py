#!/usr/bin/env python3
"""check_julia_binding.py — static check of the Julia patch in INTEGRATION.md against include/partls.h.

No Julia toolchain exists in this image, so the `ccall`s of the patch cannot be executed here.  What CAN be checked without
one: every `ccall((:sym, _PARTLS_LIB), Ret, (ArgTypes...), args...)` in the fenced ```julia blocks names a symbol the header
declares, with the same arity, a return type and argument types that are ABI-compatible with the C prototype, and as many
actual arguments as the type tuple has entries.  tests/test_julia_binding.py runs this on the CPU.

    python tools/check_julia_binding.py            # prints one line per ccall, exits 1 on the first mismatch
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# C parameter / return type (normalised: no `const`, single spaces, `*` glued) -> Julia types that are ABI-compatible with it
JULIA_FOR_C = {
    "partls_ctx*": {"Ptr{Cvoid}"},
    "partls_ctx**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "partls_multi*": {"Ptr{Cvoid}"},
    "partls_multi**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "partls_frontier*": {"Ptr{Cvoid}"},
    "partls_frontier**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "uint64_t*": {"Ptr{UInt64}", "Ref{UInt64}"},
    "int32_t*": {"Ptr{Int32}", "Ref{Int32}"},
    "int*": {"Ptr{Cint}", "Ref{Cint}", "Ptr{Int32}", "Ref{Int32}"},
    "double": {"Float64", "Cdouble"},
    "int64_t": {"Int64"},
    "uint64_t": {"UInt64"},
    "uint32_t": {"UInt32", "Cuint"},
    "int": {"Cint", "Int32"},
    "partls_timer": {"Cint", "Int32"},
    "partls_status": {"Cint", "Int32"},
    "char*": {"Cstring", "Ptr{UInt8}", "Ptr{Cchar}"},
    "void": {"Cvoid"},
}


def _split_top(s, sep=","):
    """split on `sep` at nesting depth 0 of (), [], {}"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail:
        out.append(tail)
    return out


def _norm_c(t):
    t = re.sub(r"\bconst\b", "", t)
    t = re.sub(r"\s+", " ", t).strip()
    return t.replace(" *", "*").replace("* ", "*")


def parse_header(path=None):
    """{symbol: (return type, [parameter types])} for every prototype of include/partls.h"""
    text = open(path or os.path.join(ROOT, "include", "partls.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(partls_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, params = _norm_c(m.group(1)), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        ptypes = []
        if params and params != "void":
            for p in _split_top(params):
                p = _norm_c(p)
                mm = re.match(r"^(.*?)(\**)\s*([A-Za-z_]\w*)$", p)       # type, stars, parameter name
                if not mm:
                    raise ValueError(f"{name}: cannot parse parameter '{p}'")
                ptypes.append((mm.group(1).strip() + mm.group(2)).strip())
        protos[name] = (ret, ptypes)
    return protos


def parse_ccalls(path=None):
    """[(symbol, return type, [argument types], number of actual arguments, line number)] of every ccall into _PARTLS_LIB in the
    ```julia blocks of INTEGRATION.md"""
    text = open(path or os.path.join(ROOT, "INTEGRATION.md")).read()
    calls = []
    for blk in re.finditer(r"```julia\n(.*?)```", text, flags=re.S):
        code, base = blk.group(1), text[:blk.start(1)].count("\n") + 1
        for m in re.finditer(r"ccall\(\(:(\w+),\s*_PARTLS_LIB\)\s*,", code):
            # walk to the matching parenthesis of `ccall(`
            i = m.start() + len("ccall(")
            depth, j = 1, i
            while depth:
                ch = code[j]
                depth += ch in "([{"
                depth -= ch in ")]}"
                j += 1
            parts = _split_top(code[i:j - 1])
            ret, tup, actual = parts[1], parts[2], parts[3:]
            if not (tup.startswith("(") and tup.endswith(")")):
                raise ValueError(f"ccall of {m.group(1)}: third argument is not a type tuple: {tup}")
            inner = tup[1:-1].strip()
            if inner.endswith(","):
                inner = inner[:-1]
            types = _split_top(inner) if inner else []
            calls.append((m.group(1), ret, types, len(actual), base + code[:m.start()].count("\n")))
    return calls


def check(header=None, integration=None, verbose=False):
    """Raises AssertionError on the first mismatch; returns the list of checked (symbol, line) pairs."""
    protos, calls = parse_header(header), parse_ccalls(integration)
    assert calls, "no ccall found in INTEGRATION.md"
    done = []
    for sym, ret, types, nactual, line in calls:
        where = f"INTEGRATION.md:{line} ccall :{sym}"
        assert sym in protos, f"{where}: include/partls.h declares no such symbol"
        cret, cparams = protos[sym]
        assert ret in JULIA_FOR_C[cret], f"{where}: return type {ret} does not match C '{cret}'"
        assert len(types) == len(cparams), f"{where}: {len(types)} argument types, the prototype has {len(cparams)} parameters"
        assert nactual == len(types), f"{where}: {nactual} actual arguments for {len(types)} argument types"
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert ct in JULIA_FOR_C, f"{where}: no rule for C type '{ct}' (parameter {k + 1})"
            assert jt in JULIA_FOR_C[ct], f"{where}: argument {k + 1} is {jt}, the prototype says '{ct}'"
        if verbose:
            print(f"ok  {where}  ({len(types)} arguments)")
        done.append((sym, line))
    return done


if __name__ == "__main__":
    try:
        n = len(check(verbose=True))
    except AssertionError as e:
        print("MISMATCH:", e)
        sys.exit(1)
    print(f"{n} ccalls agree with include/partls.h")
