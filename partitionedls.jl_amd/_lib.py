"""ctypes binding of include/partls.h (the same symbols the Julia shim `ccall`s)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PARTLS_LIB") or os.path.join(_HERE, "libpartls_hip.so")   # PARTLS_LIB: diagnostic builds only
CSRC = os.path.join(_HERE, "csrc")

OK, ERR_BAD_ARG, ERR_BAD_PARTITION, ERR_NONFINITE, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_CONVERGED, ERR_UNSUPPORTED, ERR_STATE, ERR_ILL_CONDITIONED = range(10)
OPT_FAITHFUL_INTERCEPT = 1
OPT_GENERIC_KERNEL = 2
T_GRAM, T_PREP, T_SWEEP, T_FINISH, T_CALIB = range(5)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i64 = C.c_int64

# every symbol include/partls.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("partls_version", C.c_int, []),
    ("partls_last_error", C.c_char_p, []),
    ("partls_device_count", C.c_int, []),
    ("partls_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("partls_destroy", None, [C.c_void_p]),
    ("partls_fit_opt", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, _i64, _i64, C.c_double,
                                 C.c_uint32, _dp, _dp, _dp, _dp, _ip, _dp]),
    ("partls_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("partls_multi_destroy", None, [C.c_void_p]),
    ("partls_multi_size", C.c_int, [C.c_void_p]),
    ("partls_multi_uses_rccl", C.c_int, [C.c_void_p]),
    ("partls_multi_context", C.c_void_p, [C.c_void_p, C.c_int]),
    ("partls_fit_opt_multi", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, _i64, _i64, C.c_double,
                                       C.c_uint32, _dp, _dp, _dp, _dp, _ip, _dp]),
    ("partls_fit_bnb_multi", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, _i64, _i64, C.c_double,
                                       _dp, _dp, _dp, _dp, _ip]),
    ("partls_multi_get_timing", C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp]),
    ("partls_opt_prepare", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_int, C.c_void_p, _i64, _i64,
                                     C.c_double, C.c_uint32]),
    ("partls_opt_sweep", C.c_int, [C.c_void_p, _i64, _i64, _dp, _ip, _dp, _ip]),
    ("partls_opt_finish", C.c_int, [C.c_void_p, _i64, _dp, _dp, _dp, _dp, _ip]),
    ("partls_opt_candidates", C.c_int, [C.c_void_p, _i64, _dp, _ip, _ip]),
    ("partls_opt_merge_candidates", C.c_int, [C.c_void_p, _i64, _dp, _ip, _dp, _ip]),
    ("partls_opt_pattern", C.c_int, [C.c_void_p, _i64, _dp, _dp]),
    ("partls_opt_num_patterns", _i64, [C.c_void_p]),
    ("partls_opt_bit_order", C.c_int, [C.c_void_p, _ip, _dp]),
    ("partls_fit_alt", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, _i64, _i64, C.c_double,
                                 C.c_double, _i64, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    ("partls_fit_bnb", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, _i64, _i64, C.c_double,
                                 _dp, _dp, _dp, _dp, _ip]),
    ("partls_alt_prepared", C.c_int, [C.c_void_p, C.c_double, _i64, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    ("partls_bnb_prepared", C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _ip]),
    ("partls_bnb_bound", C.c_int, [C.c_void_p, _i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _dp, C.POINTER(C.c_int32)]),
    ("partls_bnb_snap_begin", C.c_int, [C.c_void_p]),
    ("partls_bnb_bound_snap", C.c_int, [C.c_void_p, _i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), _dp, C.POINTER(C.c_int32)]),
    ("partls_bnb_snap_release", C.c_int, [C.c_void_p, _i64, C.POINTER(C.c_int32)]),
    ("partls_frontier_create", C.c_int, [C.c_int, C.c_int, C.c_int, _i64, C.POINTER(C.c_void_p)]),
    ("partls_frontier_destroy", None, [C.c_void_p]),
    ("partls_frontier_next", C.c_int, [C.c_void_p, _ip, _ip, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]),
    ("partls_frontier_ingest", C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i64, _ip]),
    ("partls_frontier_result", C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _ip]),
    ("partls_bnb_search", C.c_int, [C.c_void_p, _i64, _dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _ip]),
    ("partls_bnb_leaf", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _dp, _dp, _dp, _dp]),
    ("partls_predict", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, _i64, _i64, _dp, _dp, C.c_double, _dp]),
    ("partls_predict_device", C.c_int, [C.c_void_p, C.c_void_p, _i64, _i64, _i64, C.c_void_p, _i64, _i64, _dp, _dp, C.c_double,
                                        C.c_void_p]),
    ("partls_synth_truth", C.c_int, [C.c_uint64, _i64, _i64, _ip, _dp]),
    ("partls_synth_device", C.c_int, [C.c_void_p, C.c_uint64, _i64, _i64, _dp, C.c_void_p, C.c_void_p]),
    ("partls_get_timing", C.c_int, [C.c_void_p, C.c_int, _dp]),
    ("partls_get_upload", C.c_int, [C.c_void_p, _dp, _dp]),
    ("partls_get_gram", C.c_int, [C.c_void_p, _dp]),
    ("partls_get_pivots", C.c_int, [C.c_void_p, _ip]),
    ("partls_get_vetoes", C.c_int, [C.c_void_p, _ip]),
    ("partls_get_kkt_violation", C.c_int, [C.c_void_p, _dp, _dp]),
    ("partls_get_near_ties", C.c_int, [C.c_void_p, _ip]),
]

_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libpartls_hip.so (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, file name
    without the version), so a process that loads /opt/rocm's copy first and torch's second ends up with two HSA runtimes
    and torch then reports "No HIP GPUs".  When torch is installed, load ITS copy first (without importing torch): our
    DT_NEEDED libamdhip64.so.7 then binds to it by SONAME and a later `import torch` re-uses the same file."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                   # torch already brought its runtime in; SONAME matching does the rest
    if os.environ.get("PARTLS_NO_TORCH_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the HIP library. Fails loudly when it is missing: there is no fallback implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(f"{SO_PATH} is missing: build it with __graft_entry__.build() "
                              f"(make -C '{CSRC}'); partitionedls.jl_amd has no CPU fallback")
        _preload_torch_hip_runtime()
        l = C.CDLL(SO_PATH)
        for name, res, args in SYMBOLS:
            if os.environ.get("PARTLS_LIB") and not hasattr(l, name):
                continue                   # diagnostic A/B builds of older sources may lack the newest entry points
            f = getattr(l, name)           # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib
