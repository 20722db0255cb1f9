"""GPU tests of the in-library multi-device fit(Opt) (include/partls.h: partls_fit_opt_multi; Opt.jl:85-96 sharded).

A one-GPU box can exercise both halves of the design: one rank through the real RCCL communicator (ncclCommInitAll +
two ncclAllReduce(min)), and R ranks — R host threads, R contexts, R Gray-index shards — that all sit on device 0, whose
reduction runs through the host because RCCL refuses a communicator with a duplicated device.  Both must reproduce the
single-context fit: winner, objective, model, and the merged all_opt."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(oracle, seed=20260031, N=3000, D=40, K=9):
    X, y, P, _ = oracle.synth(seed, N, D, K)
    return X, y, P


def _single(partls, X, y, P, eta=0.0, want_all=False):
    L = partls.lowlevel
    ctx = partls.default_context()
    ctx.opt_prepare(X, y, P, eta, L.OPT_FAITHFUL_INTERCEPT if want_all else 0)
    bo, bp, allopt, unconv = ctx.opt_sweep(0, -1, want_all=want_all)
    assert unconv == 0
    a, b, t, opt, bi = ctx.opt_finish(bp)
    return a, b, t, opt, bi, allopt


def test_one_rank_through_rccl(partls, oracle):
    X, y, P = _problem(oracle)
    mc = partls.MultiContext([0])
    try:
        assert mc.size == 1 and mc.uses_rccl
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P)
        a1, b1, t1, opt1, bi1, _ = _single(partls, X, y, P)
        assert bi == bi1 and opt == opt1
        np.testing.assert_array_equal(a, a1)
        np.testing.assert_array_equal(b, b1)
        ref = oracle.fit_opt(X, y, P)
        assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        np.testing.assert_allclose(a, ref["alpha"], atol=1e-7)
        # a second fit on the same handle (communicator reuse), other problem
        X2, y2, P2 = _problem(oracle, seed=20260032, N=1500, D=24, K=6)
        a, b, t, opt, bi, allopt = mc.fit_opt(X2, y2, P2, want_all=True)
        ref = oracle.fit_opt(X2, y2, P2, return_all=True)
        assert bi == ref["best_index"]
        np.testing.assert_allclose(allopt, ref["all_opt"], rtol=1e-9, atol=1e-9)
    finally:
        mc.close()


@pytest.mark.parametrize("R", [2, 3, 8])
def test_R_ranks_on_one_device_equal_the_single_sweep(partls, oracle, R):
    X, y, P = _problem(oracle)
    mc = partls.MultiContext([0] * R)
    try:
        assert mc.size == R and not mc.uses_rccl
        a, b, t, opt, bi, allopt = mc.fit_opt(X, y, P, want_all=True)
        a1, b1, t1, opt1, bi1, allopt1 = _single(partls, X, y, P, want_all=True)
        assert bi == bi1 and abs(opt - opt1) <= 1e-12 * opt1
        np.testing.assert_allclose(a, a1, atol=1e-12)
        np.testing.assert_allclose(b, b1, atol=1e-12)
        assert not np.isnan(allopt).any()                             # every pattern was visited by exactly one shard
        np.testing.assert_allclose(allopt, allopt1, rtol=1e-10)       # chain paths differ per shard: objectives agree to round-off
        assert bi == int(np.argmin(allopt))
        # rank 0's context holds the problem: models of other patterns can be rebuilt on it (returnAllSolutions, Opt.jl:99-101)
        c0 = mc.context(0)
        c0._shape = (X.shape[0], X.shape[1], P.shape[1])
        assert abs(c0.opt_pattern(5)[1] - allopt[5]) <= 1e-9 * allopt[5]
        # free-intercept mode (2^K patterns over R ranks, R not a power of two for R = 3)
        a2, b2, t2, opt2, bi2, _ = mc.fit_opt(X, y, P)
        assert bi2 == bi and abs(opt2 - opt) <= 1e-12 * opt
    finally:
        mc.close()


def test_fit_api_with_devices(partls, oracle):
    """pls.fit(Opt, ..., devices=...) = what the Julia drop-in calls on a multi-GPU node; result tuple as the reference's"""
    X, y, P = _problem(oracle, seed=20260033, N=800, D=12, K=4)
    m1, _, r1 = partls.fit(partls.Opt, X, y, P)
    m2, none, r2 = partls.fit(partls.Opt, X, y, P, devices=[0, 0])
    assert none is None and r2.best_index == r1.best_index and abs(r2.opt - r1.opt) <= 1e-12 * max(1.0, r1.opt)
    np.testing.assert_allclose(m2.α, m1.α, atol=1e-12)
    m3, _, r3 = partls.fit(partls.Opt, X, y, P, devices=[0, 0], returnAllSolutions=True)
    ref = oracle.fit_opt(X, y, P, return_all=True)
    sols = list(r3.solutions)
    assert len(sols) == 1 << (P.shape[1] + 1)
    np.testing.assert_allclose([s[0] for s in sols], ref["all_opt"], rtol=1e-9, atol=1e-9)


def test_exact_tie_across_ranks_keeps_the_first_index(partls):
    """two groups without any feature: four patterns tie bitwise, in different shards — argmin's first index (Opt.jl:96) wins"""
    rng = np.random.default_rng(5)
    X = rng.standard_normal((200, 6)); y = rng.standard_normal(200)
    P = np.zeros((6, 4), dtype=np.int64); P[:3, 0] = 1; P[3:, 2] = 1           # groups 1 and 3 are empty
    mc = partls.MultiContext([0, 0, 0, 0])
    try:
        a, b, t, opt, bi, allopt = mc.fit_opt(X, y, P, want_all=True)
        m1, _, r1 = partls.fit(partls.Opt, X, y, P, faithful_intercept=True)
        assert bi == r1.best_index == int(np.argmin(allopt))
        assert (bi >> 1) & 1 == 0 and (bi >> 3) & 1 == 0
    finally:
        mc.close()


def test_errors_do_not_hang_the_ranks(partls, oracle):
    X, y, P = _problem(oracle, N=300, D=8, K=3)
    Pbad = P.copy(); Pbad[0, 0] = 2
    mc = partls.MultiContext([0, 0, 0])
    try:
        with pytest.raises(partls.PartlsError) as ei:
            mc.fit_opt(X, y, Pbad)
        assert ei.value.status == partls.lowlevel.ERR_BAD_PARTITION and "rank 0" in str(ei.value)
        Xn = X.copy(); Xn[5, 2] = np.nan
        with pytest.raises(partls.PartlsError) as ei:
            mc.fit_opt(Xn, y, P)
        assert ei.value.status == partls.lowlevel.ERR_NONFINITE
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P)                       # the handle is still usable
        assert abs(opt - oracle.fit_opt(X, y, P)["opt"]) <= 1e-9 * max(1.0, opt)
    finally:
        mc.close()


def test_raw_ctypes_multi_and_argument_checks(partls):
    lib = partls.lowlevel.lib()
    h = C.c_void_p()
    assert lib.partls_multi_create(None, 0, C.byref(h)) == partls.lowlevel.OK          # every visible device
    assert lib.partls_multi_size(h) == lib.partls_device_count() and lib.partls_multi_uses_rccl(h) == 1
    assert lib.partls_multi_context(h, lib.partls_multi_size(h)) is None
    lib.partls_multi_destroy(h)
    bad = (C.c_int * 1)(99)
    assert lib.partls_multi_create(bad, 1, C.byref(h)) == partls.lowlevel.ERR_NO_DEVICE
    assert lib.partls_multi_create(None, -1, C.byref(h)) == partls.lowlevel.ERR_BAD_ARG
    assert lib.partls_multi_size(None) == 0


def test_rows_sharded_over_the_ranks(partls, oracle, monkeypatch):
    """Every rank uploads 1/R of the rows, the Gram products of the blocks are summed (RCCL sum all-reduce; through the host in the
    one-device rehearsal), and the finish's passes over the data cover every block: same fit as with a replicated upload
    (PARTLS_MULTI_REPLICATE) and as the oracle — also with eta > 0, with a leading dimension, and when N is not a multiple of R."""
    X, y, P = _problem(oracle, seed=20260035, N=3001, D=37, K=7)
    ref = oracle.fit_opt(X, y, P)
    res = {}
    for mode in ("shard", "replicate"):
        if mode == "replicate":
            monkeypatch.setenv("PARTLS_MULTI_REPLICATE", "1")
        mc = partls.MultiContext([0, 0, 0])
        try:
            res[mode] = mc.fit_opt(X, y, P)
            # the rank-0 context holds the problem: single patterns are re-solved with data passes over all blocks
            c0 = mc.context(0); c0._shape = (X.shape[0], X.shape[1], P.shape[1])
            ra, o5 = c0.opt_finish(5)[0], c0.opt_finish(5)[3]
            res[mode + "_p5"] = o5
        finally:
            mc.close()
    a, b, t, opt, bi, _ = res["shard"]
    a2, b2, t2, opt2, bi2, _ = res["replicate"]
    assert bi == bi2 == ref["best_index"] and abs(opt - opt2) <= 1e-12 * opt2 and abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
    np.testing.assert_allclose(a, a2, atol=1e-11)
    np.testing.assert_allclose(a, ref["alpha"], atol=1e-7)
    assert abs(res["shard_p5"] - res["replicate_p5"]) <= 1e-12 * res["replicate_p5"]
    monkeypatch.delenv("PARTLS_MULTI_REPLICATE", raising=False)
    # eta > 0 (the regularisation rows are added once, not per block) and NaN in ONE rank's block
    mc = partls.MultiContext([0, 0])
    try:
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P, eta=0.7)
        refe = oracle.fit_opt(X, y, P, eta=0.7)
        assert bi == refe["best_index"] and abs(opt - refe["opt"]) <= 1e-9 * max(1.0, refe["opt"])
        Xn = X.copy(); Xn[2900, 3] = np.inf                                # in the second rank's rows
        with pytest.raises(partls.PartlsError) as ei:
            mc.fit_opt(Xn, y, P)
        assert ei.value.status == partls.lowlevel.ERR_NONFINITE
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P)                          # the handle survives
        assert bi == ref["best_index"]
    finally:
        mc.close()


# ---- round 4: near ties across shards, fit(BnB) over the ranks, fault injection ---------------------------------------------------------
def _intercept_near_tie_problem(seed, flip):
    """strong group signal, intercept ~ 0: the two patterns that differ in the intercept's sign only are the top two of the faithful
    enumeration (their objective^2 differ by ~ one unit of noise variance out of N) and sit in different halves of the Gray index"""
    rng = np.random.default_rng(seed)
    N, D, K = 400, 12, 3
    X = rng.standard_normal((N, D)); X -= X.mean(axis=0)
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    w = rng.random(D) * np.array([2.0, -3.0, 1.5])[np.arange(D) % K]
    y = X @ w + 0.5 * rng.standard_normal(N)
    y = y - y.mean() + (0.02 if flip else -0.02)
    return X, y, P


@pytest.mark.parametrize("flip", [False, True])
def test_near_tie_split_over_two_ranks_is_reranked_like_a_single_context(partls, monkeypatch, flip):
    """The finish re-ranks the winner and its near ties on the objective computed from the data (Opt.jl:90,96).  With the pattern space
    sharded, the runner-up may live on another rank: its candidates must reach the finish, so that partls_fit_opt_multi evaluates the
    set a single context would and returns the same model.  The near-tie window (PARTLS_NEAR_TIE_REL, normally 1e-13 y'y) is widened
    to sit between the gaps winner / second and winner / third of THIS problem: exactly one near tie exists, in the other shard."""
    L = partls.lowlevel
    X, y, P = _intercept_near_tie_problem(3, flip)
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
    _, _, allopt, _ = ctx.opt_sweep(0, -1, want_all=True)
    ctx.close()
    order = np.argsort(allopt, kind="stable")
    b1, b2, b3 = (int(v) for v in order[:3])
    top = P.shape[1]                                                        # the intercept's bit: the top Gray bit of a short enumeration
    assert (b1 ^ b2) == 1 << top, (b1, b2)                                  # winner and runner-up differ in the intercept's sign only
    yy = float(y @ y)
    g2, g3 = allopt[b2] ** 2 - allopt[b1] ** 2, allopt[b3] ** 2 - allopt[b1] ** 2
    assert 0 < 2.5 * g2 < g3
    monkeypatch.setenv("PARTLS_NEAR_TIE_REL", repr(float(1.6 * g2 / yy)))
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
    bo, bp, _, _ = ctx.opt_sweep(0, -1)
    co, cp = ctx.opt_candidates()
    assert list(cp) == [b1, b2] and co[0] <= co[1]
    a1, bt1, t1, opt1, bi1 = ctx.opt_finish(bp)
    assert ctx.near_ties_evaluated() == 2
    ctx.close()
    mc = partls.MultiContext([0, 0])
    try:
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P, flags=L.OPT_FAITHFUL_INTERCEPT)
        assert mc.context(0).near_ties_evaluated() == 2                      # the other shard's candidate reached rank 0's finish
        assert bi == bi1 and abs(opt - opt1) <= 1e-12 * opt1                    # (the row-sharded Gram sum rounds differently)
        np.testing.assert_allclose(a, a1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b, bt1, rtol=0, atol=1e-11)
    finally:
        mc.close()


def test_candidates_merge_abi(partls, oracle):
    """partls_opt_candidates / partls_opt_merge_candidates: what a process-per-GPU host (dist.reduce_winner) exchanges"""
    X, y, P = _problem(oracle, N=800, D=20, K=5)
    c0, c1 = partls.Context(0), partls.Context(0)
    try:
        for c in (c0, c1):
            c.opt_prepare(X, y, P)
        n = c0.num_patterns()
        o0, p0, _, _ = c0.opt_sweep(0, n // 2)
        o1, p1, _, _ = c1.opt_sweep(n // 2, n)
        lists = [c0.opt_candidates(), c1.opt_candidates()]
        assert lists[0][1][0] == p0 and lists[1][1][0] == p1
        objs = np.concatenate([l[0] for l in lists]); pats = np.concatenate([l[1] for l in lists])
        w0 = c0.opt_merge_candidates(objs, pats)
        w1 = c1.opt_merge_candidates(objs[::-1].copy(), pats[::-1].copy())      # any order
        assert w0 == w1 == min([(o0, p0), (o1, p1)])
        r0 = c0.opt_finish(w0[1]); r1 = c1.opt_finish(w1[1])
        assert abs(r0[3] - r1[3]) <= 1e-12 * r1[3] and r0[4] == r1[4]           # (one context starts from the solution its sweep left behind)
        np.testing.assert_allclose(r0[0], r1[0], rtol=0, atol=1e-12)
        ref = oracle.fit_opt(X, y, P)
        assert r0[4] == ref["best_index"] and abs(r0[3] - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        assert c0.opt_merge_candidates(np.zeros(0), np.zeros(0, dtype=np.int64)) == (float("inf"), -1)
    finally:
        c0.close(); c1.close()


def _branching(seed=7, N=400, D=36, K=6):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    y = X @ rng.standard_normal(D) + 0.2 * rng.standard_normal(N)
    return X, y, P


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_fit_bnb_multi_equals_the_single_context_search_and_the_oracle(partls, oracle, monkeypatch, devices):
    """partls_fit_bnb_multi: one rank through the real RCCL communicator (ncclAllGather per round), 2 / 3 rank threads on device 0
    through host memory — same optimum and model as partls_fit_bnb, as the oracle's depth-first recursion (BnB.jl:94-132) and as Opt.
    Small batches, so that the search takes many rounds and the dealing (owner first, surplus cold) is exercised."""
    monkeypatch.setenv("PARTLS_BNB_BATCH", "8")
    X, y, P = _branching()
    ref = oracle.fit_bnb(X, y, P)
    c1 = partls.Context(0)                                                  # (a fresh context: the batch size is read at create)
    c1.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    _, _, _, opt_single, nopen_single = c1.bnb_prepared()
    c1.close()
    mc = partls.MultiContext(devices)
    try:
        assert mc.uses_rccl == (len(devices) == 1)
        a, b, t, opt, nopen = mc.fit_bnb(X, y, P)
        assert nopen > 20
        assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and abs(opt - opt_single) <= 1e-12 * max(1.0, opt_single)
        np.testing.assert_allclose(partls.predict(partls.PartLSFitResult(a, b, t, P), X),
                                   oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]), atol=1e-7)
        if len(devices) == 1:
            assert nopen == nopen_single                                     # one rank: the very same search
        # eta > 0 and rows that do not divide by R; then the fit() front end
        a, b, t, opt, nopen = mc.fit_bnb(X[:397], y[:397], P, eta=0.3)
        refe = oracle.fit_bnb(X[:397], y[:397], P, eta=0.3)
        assert abs(opt - refe["opt"]) <= 1e-9 * max(1.0, refe["opt"])
    finally:
        mc.close()
    if len(devices) == 2:
        m, _, rep = partls.fit(partls.BnB, X, y, P, devices=devices)
        assert abs(rep.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and rep.nopen > 20


def test_fit_bnb_multi_raw_ctypes_and_errors(partls):
    lib = partls.lowlevel.lib()
    L = partls.lowlevel
    X, y, P = _branching(N=200, D=12, K=3)
    Xf = np.asfortranarray(X); Pf = np.asfortranarray(P)
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.partls_multi_create(devs, 2, C.byref(h)) == L.OK
    a = np.zeros(12); b = np.zeros(3); t = C.c_double(); o = C.c_double(); no = C.c_int64()
    dp = lambda v: v.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.partls_fit_bnb_multi(h, Xf.ctypes.data, 200, 12, 200, y.ctypes.data, Pf.ctypes.data, 3, 12, 0.0, dp(a), dp(b), C.byref(t), C.byref(o),
                                    C.byref(no)) == L.OK
    assert no.value >= 1 and o.value > 0
    assert lib.partls_fit_bnb_multi(None, Xf.ctypes.data, 200, 12, 200, y.ctypes.data, Pf.ctypes.data, 3, 12, 0.0, dp(a), dp(b), C.byref(t), C.byref(o),
                                    C.byref(no)) == L.ERR_BAD_ARG
    assert lib.partls_fit_bnb_multi(h, Xf.ctypes.data, 200, 12, 200, y.ctypes.data, Pf.ctypes.data, 3, 12, 0.0, None, dp(b), C.byref(t), C.byref(o),
                                    C.byref(no)) == L.ERR_BAD_ARG
    Pbad = Pf.copy(); Pbad[1, 1] = 3
    assert lib.partls_fit_bnb_multi(h, Xf.ctypes.data, 200, 12, 200, y.ctypes.data, Pbad.ctypes.data, 3, 12, 0.0, dp(a), dp(b), C.byref(t), C.byref(o),
                                    C.byref(no)) == L.ERR_BAD_PARTITION
    assert lib.partls_fit_bnb_multi(h, Xf.ctypes.data, 200, 12, 200, y.ctypes.data, Pf.ctypes.data, 3, 12, 0.0, dp(a), dp(b), C.byref(t), C.byref(o),
                                    C.byref(no)) == L.OK                   # the handle survives
    lib.partls_multi_destroy(h)


@pytest.mark.parametrize("alg", ["opt", "bnb"])
@pytest.mark.parametrize("stage", [1, 2, 3, 4, 5])
def test_a_rank_that_fails_at_any_stage_fails_the_fit_and_nobody_hangs(partls, oracle, monkeypatch, alg, stage):
    """Fault injection (PARTLS_MULTI_FAULT=rank:stage): rank 1 of three (rank 0 for the finish) reports an error before the upload, inside
    the Gram exchange, after its sweep / before the search, in the reduction / in the second search round, in the finish.  Every
    other rank must leave its rendezvous; the fit returns that rank's status; the handle is usable afterwards."""
    import time
    monkeypatch.setenv("PARTLS_BNB_BATCH", "8")
    monkeypatch.setenv("PARTLS_MULTI_FAULT", f"{0 if stage == 5 else 1}:{stage}")
    monkeypatch.setenv("PARTLS_MULTI_TIMEOUT_S", "20")
    X, y, P = _branching(N=300, D=18, K=4)
    mc = partls.MultiContext([0, 0, 0])
    monkeypatch.delenv("PARTLS_MULTI_FAULT")
    try:
        t0 = time.time()
        with pytest.raises(partls.PartlsError) as ei:
            mc.fit_opt(X, y, P) if alg == "opt" else mc.fit_bnb(X, y, P)
        assert time.time() - t0 < 10.0                                        # by agreement, not by timeout
        assert ei.value.status == partls.lowlevel.ERR_HIP and "injected fault" in str(ei.value), str(ei.value)
        assert f"rank {0 if stage == 5 else 1}" in str(ei.value)
    finally:
        mc.close()
    mc = partls.MultiContext([0, 0, 0])                                        # (no fault configured: the same sizes work)
    try:
        if alg == "opt":
            assert abs(mc.fit_opt(X, y, P)[3] - oracle.fit_opt(X, y, P)["opt"]) <= 1e-9
        else:
            assert abs(mc.fit_bnb(X, y, P)[3] - oracle.fit_bnb(X, y, P)["opt"]) <= 1e-9
    finally:
        mc.close()


@pytest.mark.parametrize("alg,stage", [("opt", 1), ("opt", 3), ("bnb", 3), ("bnb", 4)])   # (opt, 4: nothing is owed after the agreement in host mode)
def test_a_rank_that_vanishes_is_caught_by_the_bounded_rendezvous(partls, monkeypatch, alg, stage):
    """The rank thread simply returns (what a protocol bug would look like): the others wait PARTLS_MULTI_TIMEOUT_S at their rendezvous,
    then the fit fails with PARTLS_ERR_STATE instead of hanging; a later fit on the same handle works (the barrier is reset)."""
    import time
    monkeypatch.setenv("PARTLS_BNB_BATCH", "8")
    monkeypatch.setenv("PARTLS_MULTI_FAULT", f"1:{stage}:vanish")
    monkeypatch.setenv("PARTLS_MULTI_TIMEOUT_S", "1.5")
    X, y, P = _branching(N=300, D=18, K=4)
    mc = partls.MultiContext([0, 0])
    try:
        t0 = time.time()
        with pytest.raises(partls.PartlsError) as ei:
            mc.fit_opt(X, y, P) if alg == "opt" else mc.fit_bnb(X, y, P)
        assert 1.0 < time.time() - t0 < 15.0
        assert ei.value.status == partls.lowlevel.ERR_STATE and "rendezvous" in str(ei.value), str(ei.value)
    finally:
        mc.close()


def test_views_of_a_closed_multi_context_do_not_dangle(partls, oracle):
    X, y, P = _problem(oracle, N=400, D=10, K=3)
    mc = partls.MultiContext([0, 0])
    m, _, rep = None, None, None
    a, b, t, opt, bi, allopt = mc.fit_opt(X, y, P, want_all=True)
    view = mc.context(0)
    sols = partls.api._Solutions(view, allopt, P, (np.asfortranarray(X), y, np.asfortranarray(P), 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT, 0)) \
        if hasattr(partls, "api") else None
    mc.close()
    assert not view._h                                                       # the borrowed handle is gone with its owner
    if sols is not None:
        view._shape = (X.shape[0], X.shape[1], P.shape[1])
        o, model = sols[int(bi)]                                             # rebuilt on a private context, not through the dead view
        assert abs(o - opt) <= 1e-9 * max(1.0, opt)


def test_row_blocks_go_up_through_the_staged_upload(partls, oracle):
    """Row blocks above 8 MB are staged through page-locked buffers by copier threads INSIDE every rank thread (a strided view of the
    caller's matrix per rank): same fit as the single context, whose own upload is staged too."""
    rng = np.random.default_rng(3)
    N, D, K = 300_000, 8, 3
    X = np.asfortranarray(rng.standard_normal((N, D)))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    y = X @ (rng.random(D) * np.array([1.0, -2.0, 0.5])[np.arange(D) % K]) + 0.2 + 0.3 * rng.standard_normal(N)
    a1, b1, t1, opt1, bi1, _ = _single(partls, X, y, P)
    assert partls.default_context().upload()[1] == N * D * 8
    mc = partls.MultiContext([0, 0])
    try:
        a, b, t, opt, bi, _ = mc.fit_opt(X, y, P)
        assert mc.context(0).upload()[1] == (N // 2) * D * 8 and mc.context(1).upload()[1] == (N - N // 2) * D * 8
        assert bi == bi1 and abs(opt - opt1) <= 1e-11 * opt1
        np.testing.assert_allclose(a, a1, atol=1e-10)
        a, b, t, opt, nopen = mc.fit_bnb(X, y, P)
        assert abs(opt - opt1) <= 1e-9 * opt1
    finally:
        mc.close()
