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
