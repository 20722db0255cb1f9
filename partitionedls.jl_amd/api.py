"""Host-side mirror of the reference's operator interface for the hot path (PartitionedLS.jl:3 exports).

fit(Opt|Alt|BnB, X, y, P; η, ...) / predict keep the reference's names, argument meaning, result tuple
`(PartLSFitResult, nothing, report)` and error behaviour; the arithmetic runs on the MI355X through the C ABI
(include/partls.h).  Nothing here computes on the CPU beyond argument marshalling.
"""
import atexit
import ctypes as C
import warnings
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class PartlsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"partls status {status}: {msg}")
        self.status = status


class Opt:      # Opt.jl:1
    """Optimal algorithm: complete enumeration of the sign patterns."""


class Alt:      # Alt.jl:3
    """Alternating optimisation."""


class BnB:      # BnB.jl:1
    """Branch and bound."""


@dataclass
class PartLSFitResult:          # PartitionedLS.jl:29-49
    α: np.ndarray
    β: np.ndarray
    t: float
    P: np.ndarray

    @property
    def alpha(self):
        return self.α

    @property
    def beta(self):
        return self.β


class Report(dict):
    """The NamedTuple third element of fit's result: .opt (all), .nopen (BnB), .solutions (Opt, returnAllSolutions)."""
    __getattr__ = dict.__getitem__


def library_path():
    return L.SO_PATH


def build_library(force=False):
    return L.build(force)


def _check(st, tolerate=()):
    """Raise on a non-zero status, except one the caller handles itself (it gets the status back)."""
    if st != L.OK and st not in tolerate:
        raise PartlsError(st, L.lib().partls_last_error().decode())
    return st


class IllConditionedWarning(UserWarning):
    """The returned model failed the data-space KKT check (PARTLS_ERR_ILL_CONDITIONED): the fp64 Gram form cannot resolve this X."""


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Context:
    """Thin owner of a partls_ctx (one per device)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(L.lib().partls_create(int(device), C.byref(self._h)))
        self.device = device
        self.generation = 0          # bumped by every prepare: lazily rebuilt results check it (see _Solutions)
        self.tolerate_ill = False    # True: status 9 (outputs hold the best Gram-form model) is recorded in last_ill instead of raised
        self.last_ill = False
        _live_contexts.add(self)

    def _ill(self, st):
        """status of a call whose outputs are filled even when it reports PARTLS_ERR_ILL_CONDITIONED (include/partls.h)"""
        _check(st, (L.ERR_ILL_CONDITIONED,) if getattr(self, "tolerate_ill", False) else ())
        self.last_ill = st == L.ERR_ILL_CONDITIONED

    def close(self):
        """Release the device objects now.  Also run for every live context by an atexit hook, i.e. BEFORE interpreter
        finalisation and before the HIP runtime (or a profiler layered on it) tears itself down — a context destroyed later,
        from a module-global's __del__ during exit(), made HIP calls into a dead runtime (round-1 rocprofv3 crash)."""
        if self._h and not getattr(self, "_borrowed", False):      # a view of a MultiContext's rank does not own the handle
            L.lib().partls_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- staged Opt path -------------------------------------------------------------------------------------------
    def opt_prepare(self, X, y, P, eta=0.0, flags=0):
        """X, y: host arrays (float64, any layout; copied F-contiguous)."""
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        P = np.asfortranarray(P, dtype=np.int64)
        N, M = X.shape
        self._shape = (N, M, P.shape[1])
        self.generation += 1
        _check(L.lib().partls_opt_prepare(self._h, X.ctypes.data, N, M, N, y.ctypes.data, 0, P.ctypes.data, P.shape[1],
                                          P.shape[0], float(eta), int(flags)))

    def opt_prepare_device(self, dX_ptr, dy_ptr, N, M, ldX, P, eta=0.0, flags=0):
        """dX_ptr, dy_ptr: raw device addresses (e.g. torch tensor .data_ptr()) that stay owned by the caller."""
        P = np.asfortranarray(P, dtype=np.int64)
        self._shape = (N, M, P.shape[1])
        self.generation += 1
        _check(L.lib().partls_opt_prepare(self._h, C.c_void_p(dX_ptr), N, M, ldX, C.c_void_p(dy_ptr), 1, P.ctypes.data,
                                          P.shape[1], P.shape[0], float(eta), int(flags)))

    def num_patterns(self):
        return int(L.lib().partls_opt_num_patterns(self._h))

    def bit_order(self):
        """(gbit, flip_cost): gbit[k] = Gray-index bit that carries group k in this context's sweeps; flip_cost[k] = measured
        pivots per flip of group k (-1 where the calibration did not run).  Runs the calibration if no sweep has yet."""
        kb = self.num_patterns().bit_length() - 1
        g = np.zeros(kb, dtype=np.int64)
        fc = np.zeros(kb)
        _check(L.lib().partls_opt_bit_order(self._h, _ip(g), _dp(fc)))
        return g, fc

    def opt_sweep(self, g_begin=0, g_end=-1, want_all=False):
        bo = C.c_double()
        bp = C.c_int64()
        nu = C.c_int64()
        allopt = np.full(self.num_patterns(), np.nan) if want_all else None
        _check(L.lib().partls_opt_sweep(self._h, int(g_begin), int(g_end), C.byref(bo), C.byref(bp),
                                        _dp(allopt) if want_all else None, C.byref(nu)))
        return bo.value, bp.value, allopt, nu.value

    def opt_finish(self, pattern):
        N, M, K = self._shape
        a = np.zeros(M)
        b = np.zeros(K)
        t = C.c_double()
        o = C.c_double()
        bi = C.c_int64()
        self._ill(L.lib().partls_opt_finish(self._h, int(pattern), _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(bi)))
        return a, b, t.value, o.value, bi.value

    def opt_candidates(self):
        """(obj, pattern) arrays: this context's winner of the last sweep and its near ties (tracked objectives), best first."""
        o = np.zeros(4); p = np.zeros(4, dtype=np.int64); n = C.c_int64()
        _check(L.lib().partls_opt_candidates(self._h, 4, _dp(o), _ip(p), C.byref(n)))
        return o[:n.value].copy(), p[:n.value].copy()

    def opt_merge_candidates(self, objs, pats):
        """Install the merged candidate lists of ALL ranks (same list on every rank): returns the global (objective, pattern) winner;
        the next opt_finish(pattern) re-ranks it against its near ties on the data objective, as a single context would."""
        o = np.ascontiguousarray(objs, dtype=np.float64); p = np.ascontiguousarray(pats, dtype=np.int64)
        wo = C.c_double(); wp = C.c_int64()
        _check(L.lib().partls_opt_merge_candidates(self._h, len(o), _dp(o), _ip(p), C.byref(wo), C.byref(wp)))
        return wo.value, wp.value

    def near_ties_evaluated(self):
        n = C.c_int64()
        _check(L.lib().partls_get_near_ties(self._h, C.byref(n)))
        return n.value

    def opt_pattern(self, pattern):
        N, M, K = self._shape
        ra = np.zeros(M + 1)
        o = C.c_double()
        _check(L.lib().partls_opt_pattern(self._h, int(pattern), _dp(ra), C.byref(o)))
        return ra, o.value

    def alt_prepared(self, alpha0, beta0, eps=1e-6, T=100):
        """Alt on a context prepared with OPT_FAITHFUL_INTERCEPT (e.g. device-resident inputs)."""
        N, M, K = self._shape
        a0 = np.ascontiguousarray(alpha0, dtype=np.float64); b0 = np.ascontiguousarray(beta0, dtype=np.float64)
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); it = C.c_int64()
        self._ill(L.lib().partls_alt_prepared(self._h, float(eps), int(T), _dp(a0), _dp(b0), _dp(a), _dp(b), C.byref(t),
                                              C.byref(o), C.byref(it)))
        return a, b, t.value, o.value, it.value

    def bnb_prepared(self):
        N, M, K = self._shape
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); no = C.c_int64()
        self._ill(L.lib().partls_bnb_prepared(self._h, _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(no)))
        return a, b, t.value, o.value, no.value

    def bnb_bound(self, pats, frees):
        """Bound a batch of BnB nodes (pat[i], free[i]) on a prepared faithful context: (lb[count], branch[count])."""
        pats = np.ascontiguousarray(pats, dtype=np.uint64)
        frees = np.ascontiguousarray(frees, dtype=np.uint64)
        n = len(pats)
        lb = np.zeros(n)
        br = np.zeros(n, dtype=np.int32)
        if n:
            _check(L.lib().partls_bnb_bound(self._h, n, pats.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            frees.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(lb),
                                            br.ctypes.data_as(C.POINTER(C.c_int32))))
        return lb, br

    def bnb_snap_begin(self):
        _check(L.lib().partls_bnb_snap_begin(self._h))

    def bnb_bound_snap(self, pats, frees, src_slots):
        """bnb_bound with tableau snapshots: node i starts from slot src_slots[i] (-1: fresh tableau); returns (lb, branch, dst_slots)."""
        pats = np.ascontiguousarray(pats, dtype=np.uint64)
        frees = np.ascontiguousarray(frees, dtype=np.uint64)
        src = np.ascontiguousarray(src_slots, dtype=np.int32)
        n = len(pats)
        lb = np.zeros(n); br = np.zeros(n, dtype=np.int32); dst = np.full(n, -1, dtype=np.int32)
        if n:
            i32 = C.POINTER(C.c_int32)
            _check(L.lib().partls_bnb_bound_snap(self._h, n, pats.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                 frees.ctypes.data_as(C.POINTER(C.c_uint64)), src.ctypes.data_as(i32),
                                                 dst.ctypes.data_as(i32), _dp(lb), br.ctypes.data_as(i32)))
        return lb, br, dst

    def bnb_snap_release(self, slots):
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        if len(sl):
            _check(L.lib().partls_bnb_snap_release(self._h, len(sl), sl.ctypes.data_as(C.POINTER(C.c_int32))))

    def bnb_search(self, max_nodes=0):
        """The BnB search on a prepared faithful context (warm-started node bounds): (mu, pat, free, nodes_bounded)."""
        mu = C.c_double(); pat = C.c_uint64(); fr = C.c_uint64(); nn = C.c_int64()
        _check(L.lib().partls_bnb_search(self._h, int(max_nodes), C.byref(mu), C.byref(pat), C.byref(fr), C.byref(nn)))
        return mu.value, pat.value, fr.value, nn.value

    def bnb_leaf(self, pat, free):
        N, M, K = self._shape
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double()
        self._ill(L.lib().partls_bnb_leaf(self._h, C.c_uint64(int(pat)), C.c_uint64(int(free)), _dp(a), _dp(b), C.byref(t), C.byref(o)))
        return a, b, t.value, o.value

    def timing(self, which):
        ms = C.c_double()
        _check(L.lib().partls_get_timing(self._h, int(which), C.byref(ms)))
        return ms.value

    def upload(self):
        """(ms, bytes) of the host -> device upload of X inside the last prepare / fit (0, 0: device-resident inputs)"""
        ms = C.c_double(); b = C.c_double()
        _check(L.lib().partls_get_upload(self._h, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    def pivots(self):
        n = C.c_int64()
        _check(L.lib().partls_get_pivots(self._h, C.byref(n)))
        return n.value

    def vetoes(self):
        n = C.c_int64()
        _check(L.lib().partls_get_vetoes(self._h, C.byref(n)))
        return n.value

    def kkt_violation(self):
        """data-space KKT violation of the last finished winner (include/partls.h: partls_get_kkt_violation)"""
        v = C.c_double()
        _check(L.lib().partls_get_kkt_violation(self._h, C.byref(v), None))
        return v.value

    def min_pivot(self):
        """smallest leave-one-out pivot of the last finished model's basis (lower bound of 1 / cond of its Gram block)"""
        v = C.c_double(); mp = C.c_double()
        _check(L.lib().partls_get_kkt_violation(self._h, C.byref(v), C.byref(mp)))
        return mp.value

    def gram(self):
        N, M, K = self._shape
        G = np.zeros((M + 2, M + 2), order="F")
        _check(L.lib().partls_get_gram(self._h, _dp(G)))
        return G

    def synth_device(self, seed, N, D, wstar, dX_ptr, dy_ptr):
        ws = np.ascontiguousarray(wstar, dtype=np.float64)
        _check(L.lib().partls_synth_device(self._h, C.c_uint64(seed), N, D, _dp(ws), C.c_void_p(dX_ptr), C.c_void_p(dy_ptr)))


class Frontier:
    """The frontier of the BnB search as a native object (include/partls.h: partls_frontier_*): pops and deals rounds of nodes,
    ingests the ranks' (bound, branch, snapshot slot) triples, keeps the snapshot reference counts.  dist.bnb_search_warm drives it."""

    def __init__(self, n_groups, rank=0, world=1, batch=1024):
        self._h = C.c_void_p()
        _check(L.lib().partls_frontier_create(int(n_groups), int(rank), int(world), int(batch), C.byref(self._h)))
        self.world, self.batch = int(world), int(batch)
        self._pat = np.zeros(self.batch, dtype=np.uint64); self._free = np.zeros(self.batch, dtype=np.uint64)
        self._src = np.zeros(self.batch, dtype=np.int32); self._per = np.zeros(self.world, dtype=np.int32)
        self._dead = np.zeros(4 * self.batch * self.world + 64, dtype=np.int32)

    def close(self):
        if self._h:
            L.lib().partls_frontier_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def next(self):
        """(total, pats, frees, src_slots, per_rank): this rank's share of the next round (views valid until the next call)"""
        tot = C.c_int64(); mine = C.c_int64()
        u64 = C.POINTER(C.c_uint64); i32 = C.POINTER(C.c_int32)
        _check(L.lib().partls_frontier_next(self._h, C.byref(tot), C.byref(mine), self._pat.ctypes.data_as(u64), self._free.ctypes.data_as(u64),
                                            self._src.ctypes.data_as(i32), self._per.ctypes.data_as(i32)))
        m = mine.value
        return tot.value, self._pat[:m], self._free[:m], self._src[:m], self._per.copy()

    def ingest(self, lb, branch, dst):
        """results of the whole round in rank-major order; returns this rank's dead snapshot slots"""
        lb = np.ascontiguousarray(lb, dtype=np.float64); br = np.ascontiguousarray(branch, dtype=np.int32)
        ds = np.ascontiguousarray(dst, dtype=np.int32)
        nd = C.c_int64()
        i32 = C.POINTER(C.c_int32)
        _check(L.lib().partls_frontier_ingest(self._h, _dp(lb), br.ctypes.data_as(i32), ds.ctypes.data_as(i32),
                                              self._dead.ctypes.data_as(i32), len(self._dead), C.byref(nd)))
        return self._dead[:nd.value].copy()

    def result(self):
        mu = C.c_double(); pat = C.c_uint64(); fr = C.c_uint64(); nn = C.c_int64()
        _check(L.lib().partls_frontier_result(self._h, C.byref(mu), C.byref(pat), C.byref(fr), C.byref(nn)))
        return mu.value, pat.value, fr.value, nn.value


class MultiContext:
    """Owner of a partls_multi: fit(Opt) sharded over several GPUs of one node inside the library (one host thread and one
    context per device, RCCL min-reduce; include/partls.h: partls_fit_opt_multi).  devices: None = every visible device, an int
    n = devices 0..n-1, or a list of device indices (a list naming a device twice rehearses the R-rank control flow on one GPU:
    its reduction then runs through the host)."""

    def __init__(self, devices=None):
        self._h = C.c_void_p()
        if devices is None:
            arr, n = None, 0
        elif isinstance(devices, (int, np.integer)):
            arr, n = None, int(devices)
        else:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            n = len(devices)
        _check(L.lib().partls_multi_create(arr, n, C.byref(self._h)))
        self.generation = 0          # bumped by every fit (see _Solutions)
        self.devices = [int(d) for d in devices] if arr is not None else list(range(self.size))
        self.tolerate_ill = False
        self.last_ill = False
        self._views = weakref.WeakSet()
        _live_contexts.add(self)

    def close(self):
        if self._h:
            # the rank contexts die with the handle: views of them must not keep the raw pointers (a _Solutions object built on one
            # notices the generation change and prepares its problem again on a context of its own)
            self.generation += 1
            for v in list(self._views):
                v._h = C.c_void_p()
            L.lib().partls_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return int(L.lib().partls_multi_size(self._h))

    @property
    def uses_rccl(self):
        return bool(L.lib().partls_multi_uses_rccl(self._h))

    def context(self, rank=0):
        """Non-owning view of rank r's context (rank 0 holds the fitted problem: opt_finish / opt_pattern work on it)."""
        h = L.lib().partls_multi_context(self._h, int(rank))
        if not h:
            raise IndexError("rank out of range")
        view = Context.__new__(Context)
        view._h, view.device, view.generation, view._borrowed, view._owner = C.c_void_p(h), None, 0, True, self
        view.tolerate_ill, view.last_ill = self.tolerate_ill, False
        self._views.add(view)
        return view

    def fit_opt(self, X, y, P, eta=0.0, flags=0, want_all=False):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        P = np.asfortranarray(P, dtype=np.int64)
        N, M = X.shape
        K = P.shape[1]
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); bi = C.c_int64()
        allopt = np.full(1 << (K + 1), np.nan) if want_all else None
        self.generation += 1
        st = _check(L.lib().partls_fit_opt_multi(self._h, X.ctypes.data, N, M, N, y.ctypes.data, P.ctypes.data, K, M, float(eta),
                                                 int(flags), _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(bi),
                                                 _dp(allopt) if want_all else None),
                    (L.ERR_ILL_CONDITIONED,) if self.tolerate_ill else ())
        self.last_ill = st == L.ERR_ILL_CONDITIONED
        self._shape = (N, M, K)
        return a, b, t.value, o.value, bi.value, allopt

    def fit_bnb(self, X, y, P, eta=0.0):
        """fit(BnB) with the frontier search sharded over the ranks (include/partls.h: partls_fit_bnb_multi): (alpha, beta, t, opt, nopen)"""
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        P = np.asfortranarray(P, dtype=np.int64)
        N, M = X.shape
        K = P.shape[1]
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); no = C.c_int64()
        self.generation += 1
        st = _check(L.lib().partls_fit_bnb_multi(self._h, X.ctypes.data, N, M, N, y.ctypes.data, P.ctypes.data, K, M, float(eta),
                                                 _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(no)),
                    (L.ERR_ILL_CONDITIONED,) if self.tolerate_ill else ())
        self.last_ill = st == L.ERR_ILL_CONDITIONED
        self._shape = (N, M, K)
        return a, b, t.value, o.value, no.value

    def timing(self, rank, which):
        ms = C.c_double()
        _check(L.lib().partls_multi_get_timing(self._h, int(rank), int(which), C.byref(ms)))
        return ms.value


def synth_truth(seed, D, K):
    """Partition matrix and true weights of the BASELINE.md §4 synthetic problem (host side, tiny)."""
    P = np.zeros((D, K), dtype=np.int64, order="F")
    ws = np.zeros(D)
    _check(L.lib().partls_synth_truth(C.c_uint64(seed), D, K, _ip(P), _dp(ws)))
    return P, ws


_default_ctx = {}
_default_multi = {}
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_all_contexts():
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass
    _default_ctx.clear()
    _default_multi.clear()


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def default_multi(devices=None):
    key = devices if devices is None or isinstance(devices, (int, np.integer)) else tuple(int(d) for d in devices)
    if key not in _default_multi:
        _default_multi[key] = MultiContext(devices)
    return _default_multi[key]


# ---------------------------------------------------------------------------------------------------------------------
# L2 helpers the reference exports (PartitionedLS.jl:76-81, :108-123) — shape bookkeeping only, kept for API parity
# ---------------------------------------------------------------------------------------------------------------------
def homogeneousCoords(X, P):
    X = np.asarray(X)
    P = np.asarray(P, dtype=np.int64)
    Xo = np.hstack([X, np.ones((X.shape[0], 1), dtype=X.dtype)])
    Po = np.zeros((P.shape[0] + 1, P.shape[1] + 1), dtype=np.int64)
    Po[:-1, :-1] = P
    Po[-1, -1] = 1
    return Xo, Po


def regularizeProblem(X, y, P, η):
    if η == 0:
        return X, y
    X = np.asarray(X)
    rows = [np.sqrt(η) * (np.asarray(P)[:, k] == 1).astype(X.dtype)[None, :] for k in range(np.asarray(P).shape[1])]
    return np.vstack([X] + rows), np.concatenate([np.asarray(y), np.zeros(len(rows), dtype=X.dtype)])


# ---------------------------------------------------------------------------------------------------------------------
# fit / predict
# ---------------------------------------------------------------------------------------------------------------------
def _marshal(X, y, P):
    X = np.asarray(X)
    y = np.asarray(y)
    P = np.asarray(P)
    if X.ndim != 2 or y.ndim != 1:
        raise TypeError("fit: X must be a matrix and y a vector (MethodError in the reference)")
    if not (np.issubdtype(X.dtype, np.floating) and np.issubdtype(y.dtype, np.floating)):
        raise TypeError("fit: X and y must be floating point (X::Array{<:AbstractFloat,2}, Opt.jl:73)")
    if not np.issubdtype(P.dtype, np.integer) or P.ndim != 2:
        raise TypeError("fit: P must be an integer matrix (P::Array{Int,2})")
    if X.shape[0] != y.shape[0] or P.shape[0] != X.shape[1]:
        raise ValueError("DimensionMismatch: X is %s, y is %s, P is %s" % (X.shape, y.shape, P.shape))
    # Float32 inputs (test/runtests.jl:123-146) are widened; result fields are abstract floats in the reference
    return (np.asfortranarray(X, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64),
            np.asfortranarray(P, dtype=np.int64))


class _Solutions:
    """returnAllSolutions (Opt.jl:99-101): element b is (opt_b, PartLSFitResult_b); models are rebuilt on demand from the context
    that holds the fitted problem.  They stay valid whatever is fitted afterwards, as in the reference: when a later fit has taken
    the shared context over, the problem is prepared once more on a private context owned by this object."""

    def __init__(self, ctx, all_opt, P, problem):
        self._ctx, self._all, self._P, self._problem = ctx, all_opt, P, problem
        self._owner = getattr(ctx, "_owner", ctx)      # a view of a MultiContext's rank 0: the owner counts the fits
        self._gen = self._owner.generation

    def __len__(self):
        return len(self._all)

    def _context(self):
        if self._owner.generation != self._gen:        # the shared context now holds another problem
            Xf, yf, Pf, eta, flags, device = self._problem
            self._ctx = self._owner = Context(device)
            self._ctx.opt_prepare(Xf, yf, Pf, eta, flags)
            self._gen = self._ctx.generation
        return self._ctx

    def __getitem__(self, b):
        if b < 0:
            b += len(self)
        a, bt, t, opt, _ = self._context().opt_finish(b)
        return float(self._all[b]), PartLSFitResult(a, bt, t, self._P)

    def __iter__(self):
        return (self[b] for b in range(len(self)))


def fit(alg, X, y, P, *, η=None, eta=None, ϵ=None, eps=None, T=100, nnlsalg="nnls", returnAllSolutions=False, rng=None,
        alpha0=None, beta0=None, device=0, devices=None, faithful_intercept=False, generic_kernel=False, on_ill_conditioned="warn"):
    """fit(::Type{Opt|Alt|BnB}, X, y, P; η, ...) -> (PartLSFitResult, None, Report)   [Opt.jl:73, Alt.jl:50, BnB.jl:30]

    η/eta: regularisation (default 0.0);  Alt: ϵ/eps (1e-6), T (100), rng (None | int seed | numpy Generator) or an
    explicit starting point alpha0[M+1], beta0[K+1];  Opt: returnAllSolutions.
    faithful_intercept=True enumerates the reference's 2^(K+1) patterns instead of 2^K with a free intercept
    (same optimum).  Opt / BnB, devices=... (None | count | list of device indices): the enumeration / the frontier search is sharded
    over those GPUs inside the library (partls_fit_opt_multi: RCCL min-reduce; partls_fit_bnb_multi: one all-gather per round), as the
    Julia drop-in does on a multi-GPU node.  nnlsalg is accepted for signature parity; the device solver is an exact active-set method.
    on_ill_conditioned: what to do when the returned model fails the data-space KKT check (status 9: X beyond the fp64 Gram form;
    the reference's QR-based NNLS still returns a model there, and the Julia patch reroutes to it): "warn" (default) returns the best
    Gram-form model with report.ill_conditioned = True and report.kkt_violation set, and emits IllConditionedWarning; "raise" raises
    PartlsError(status 9).
    """
    if alg not in (Opt, Alt, BnB):
        raise TypeError("fit: first argument must be Opt, Alt or BnB")
    eta_v = 0.0 if (η is None and eta is None) else float(η if η is not None else eta)
    eps_v = 1e-6 if (ϵ is None and eps is None) else float(ϵ if ϵ is not None else eps)
    if nnlsalg not in ("nnls", "pivot", "fnnls"):
        raise ValueError("nnlsalg must be one of :nnls, :pivot, :fnnls")
    if on_ill_conditioned not in ("warn", "raise"):
        raise ValueError('on_ill_conditioned must be "warn" or "raise"')
    Xf, yf, Pf = _marshal(X, y, P)
    N, M = Xf.shape
    K = Pf.shape[1]
    ctx = default_context(device)
    ctx.tolerate_ill = on_ill_conditioned == "warn"
    ctx.last_ill = False
    lib = L.lib()
    Pout = np.array(Pf, dtype=np.int64, order="C")

    def report(owner, **kw):
        """the NamedTuple of the reference + what the data-space check said"""
        if owner.last_ill:
            kkt = ctx.kkt_violation() if owner is ctx else float("nan")
            warnings.warn("partitionedls: the model's KKT conditions do not hold in data space (X is too ill-conditioned for the fp64 "
                          "Gram form); the returned model is the best Gram-form one — " + L.lib().partls_last_error().decode(),
                          IllConditionedWarning, stacklevel=3)
            kw.update(ill_conditioned=True, kkt_violation=kkt)
        return Report(**kw)

    if alg is Opt:
        flags = (L.OPT_FAITHFUL_INTERCEPT if (faithful_intercept or returnAllSolutions) else 0) | \
                (L.OPT_GENERIC_KERNEL if generic_kernel else 0)
        if devices is not None:
            mc = default_multi(devices)
            mc.tolerate_ill = ctx.tolerate_ill
            a, b, t, opt, bi, allopt = mc.fit_opt(Xf, yf, Pf, eta_v, flags, want_all=returnAllSolutions)
            model = PartLSFitResult(a, b, t, Pout)
            if returnAllSolutions:
                c0 = mc.context(0)
                c0._shape = (N, M, K)
                return model, None, report(mc, solutions=_Solutions(c0, allopt, Pout, (Xf, yf, Pf, eta_v, flags, mc.devices[0])))
            return model, None, report(mc, opt=opt, best_index=bi)
        ctx.opt_prepare(Xf, yf, Pf, eta_v, flags)
        bobj, bpat, allopt, unconv = ctx.opt_sweep(0, -1, want_all=returnAllSolutions)
        if unconv:
            raise PartlsError(L.ERR_NOT_CONVERGED, f"{unconv} subproblems hit the pivot cap")
        a, b, t, opt, bi = ctx.opt_finish(bpat)
        model = PartLSFitResult(a, b, t, Pout)
        if returnAllSolutions:
            return model, None, report(ctx, solutions=_Solutions(ctx, allopt, Pout, (Xf, yf, Pf, eta_v, flags, device)))
        return model, None, report(ctx, opt=opt, best_index=bi)
    if alg is Alt:
        if alpha0 is None or beta0 is None:
            if rng is None:
                gen = np.random.default_rng()
            elif isinstance(rng, (int, np.integer)):
                gen = np.random.default_rng(int(rng))
            else:
                gen = rng
            alpha0 = gen.random(M + 1)                    # Alt.jl:65
            beta0 = (gen.random(K + 1) - 0.5) * 10        # Alt.jl:66
        a0 = np.ascontiguousarray(alpha0, dtype=np.float64)
        b0 = np.ascontiguousarray(beta0, dtype=np.float64)
        if a0.shape != (M + 1,) or b0.shape != (K + 1,):
            raise ValueError("alpha0 must have M+1 and beta0 K+1 entries")
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); it = C.c_int64()
        ctx.generation += 1
        ctx._ill(lib.partls_fit_alt(ctx._h, Xf.ctypes.data, N, M, N, yf.ctypes.data, Pf.ctypes.data, K, M, eta_v, eps_v,
                                    int(T), _dp(a0), _dp(b0), _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(it)))
        return PartLSFitResult(a, b, t.value, Pout), None, report(ctx, opt=o.value, iters=it.value)
    if alg is BnB:
        if devices is not None:
            mc = default_multi(devices)
            mc.tolerate_ill = ctx.tolerate_ill
            a, b, t, opt, nopen = mc.fit_bnb(Xf, yf, Pf, eta_v)
            return PartLSFitResult(a, b, t, Pout), None, report(mc, opt=opt, nopen=nopen)
        a = np.zeros(M); b = np.zeros(K)
        t = C.c_double(); o = C.c_double(); no = C.c_int64()
        ctx.generation += 1
        ctx._ill(lib.partls_fit_bnb(ctx._h, Xf.ctypes.data, N, M, N, yf.ctypes.data, Pf.ctypes.data, K, M, eta_v,
                                    _dp(a), _dp(b), C.byref(t), C.byref(o), C.byref(no)))
        return PartLSFitResult(a, b, t.value, Pout), None, report(ctx, opt=o.value, nopen=no.value)
    raise TypeError("fit: first argument must be Opt, Alt or BnB")


def predict(*args, device=0):
    """predict(model, X) or predict(α, β, t, P, X)  ->  X * (P .* α) * β .+ t     [PartitionedLS.jl:132-134,152-155]"""
    if len(args) == 2:
        model, X = args
        α, β, t, P = model.α, model.β, model.t, model.P
    elif len(args) == 5:
        α, β, t, P, X = args
    else:
        raise TypeError("predict(model, X) or predict(α, β, t, P, X)")
    X = np.asarray(X)
    if not np.issubdtype(X.dtype, np.floating) or X.ndim != 2:
        raise TypeError("predict: X must be a floating-point matrix")
    Xf = np.asfortranarray(X, dtype=np.float64)
    Pf = np.asfortranarray(P, dtype=np.int64)
    a = np.ascontiguousarray(α, dtype=np.float64)
    b = np.ascontiguousarray(β, dtype=np.float64)
    N, M = Xf.shape
    if Pf.shape[0] != M or a.shape != (M,) or b.shape != (Pf.shape[1],):
        raise ValueError("DimensionMismatch in predict")
    yh = np.zeros(N)
    ctx = default_context(device)
    _check(L.lib().partls_predict(ctx._h, Xf.ctypes.data, N, M, N, Pf.ctypes.data, Pf.shape[1], M, _dp(a), _dp(b),
                                  float(t), _dp(yh)))
    return yh


def predict_device(model, dX_ptr, N, ldX, dyhat_ptr, device=0):
    """predict with X (N x M, column-major, leading dimension ldX) and yhat (N) resident in HBM: raw device addresses (e.g.
    torch tensor .data_ptr()) that stay owned by the caller.  PartitionedLS.jl:132-134 without the PCIe copy of X."""
    Pf = np.asfortranarray(model.P, dtype=np.int64)
    a = np.ascontiguousarray(model.α, dtype=np.float64)
    b = np.ascontiguousarray(model.β, dtype=np.float64)
    M, K = Pf.shape
    ctx = default_context(device)
    _check(L.lib().partls_predict_device(ctx._h, C.c_void_p(dX_ptr), int(N), M, int(ldX), Pf.ctypes.data, K, M, _dp(a), _dp(b),
                                         float(model.t), C.c_void_p(dyhat_ptr)))
