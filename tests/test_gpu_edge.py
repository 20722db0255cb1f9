"""GPU edge cases (the reference tests none of these; the oracle is the checker).  Where the minimiser is not unique
(rank-deficient designs) only the objective is compared — parity is on the optimum of each convex subproblem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _check(partls, oracle, X, y, P, eta=0.0, model=True, algs=("Opt", "BnB")):
    ref = oracle.fit_opt(X, y, P, eta=eta, return_all=True)
    for name in algs:
        alg = getattr(partls, name)
        for faithful in ((True, False) if name == "Opt" else (False,)):
            kw = dict(faithful_intercept=faithful) if name == "Opt" else {}
            m, _, rep = partls.fit(alg, X, y, P, η=eta, **kw)
            assert abs(rep.opt - ref["opt"]) <= TOL * max(1.0, ref["opt"]), (name, faithful, rep.opt, ref["opt"])
            yhat = partls.predict(m, X)
            if model:
                np.testing.assert_allclose(m.α, ref["alpha"], atol=1e-7)
                np.testing.assert_allclose(m.β, ref["beta"], atol=1e-7)
            # the fitted function reproduces the objective (η = 0): ||ŷ - y|| == opt
            if eta == 0.0 and np.all(np.isfinite(yhat)):
                assert abs(np.linalg.norm(yhat - y) - ref["opt"]) <= 1e-7 * max(1.0, ref["opt"])
    return ref


def test_single_group_is_plain_least_squares_with_sign(partls, oracle):
    """MLJ's empty-P fallback (PartitionedLS.jl:318-322): one all-ones group."""
    rng = np.random.default_rng(1)
    X = rng.standard_normal((200, 6)); y = X @ np.array([1., 2, 0.5, 0.1, 3, 1]) + 0.3 + 0.05 * rng.standard_normal(200)
    _check(partls, oracle, X, y, np.ones((6, 1), dtype=np.int64))


def test_one_feature(partls, oracle):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((50, 1)); y = -2.0 * X[:, 0] + 1.0
    _check(partls, oracle, X, y, np.ones((1, 1), dtype=np.int64))


def test_every_feature_its_own_group(partls, oracle):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((300, 7)); y = X @ rng.standard_normal(7) - 0.7 + 0.1 * rng.standard_normal(300)
    ref = _check(partls, oracle, X, y, np.eye(7, dtype=np.int64))
    # with singleton groups the optimum is the unconstrained least-squares fit
    Xo = np.hstack([X, np.ones((300, 1))])
    assert abs(ref["opt"] - np.linalg.norm(Xo @ np.linalg.lstsq(Xo, y, rcond=None)[0] - y)) < 1e-9


def test_underdetermined_rank_deficient(partls, oracle):
    """N < M+1: the Gram matrix is singular; objective parity only (minimiser not unique)."""
    rng = np.random.default_rng(4)
    X = rng.standard_normal((8, 12)); y = rng.standard_normal(8)
    P = np.zeros((12, 3), dtype=np.int64); P[np.arange(12), np.arange(12) % 3] = 1
    _check(partls, oracle, X, y, P, model=False)


def test_duplicate_and_zero_columns(partls, oracle):
    rng = np.random.default_rng(5)
    X = rng.standard_normal((120, 8))
    X[:, 3] = X[:, 1]            # exact duplicate inside one group
    X[:, 6] = 0.0                # null feature
    y = X @ np.array([1., 1, -2, 1, -1, 0.5, 9, 2]) + 0.2 + 0.1 * rng.standard_normal(120)
    P = np.zeros((8, 2), dtype=np.int64); P[:4, 0] = 1; P[4:, 1] = 1
    _check(partls, oracle, X, y, P, model=False)


def test_empty_group_and_unassigned_feature(partls, oracle):
    rng = np.random.default_rng(6)
    X = rng.standard_normal((150, 5)); y = X[:, 0] - X[:, 3] + 0.1 * rng.standard_normal(150)
    P = np.zeros((5, 3), dtype=np.int64); P[0, 0] = P[1, 0] = 1; P[3, 2] = P[4, 2] = 1     # group 1 empty, feature 2 in no group
    _check(partls, oracle, X, y, P, algs=("Opt",))


def test_zero_target(partls, oracle):
    rng = np.random.default_rng(7)
    X = rng.standard_normal((60, 4)); y = np.zeros(60)
    P = np.array([[1, 0], [1, 0], [0, 1], [0, 1]], dtype=np.int64)
    m, _, rep = partls.fit(partls.Opt, X, y, P)
    assert rep.opt == 0.0 and np.all(m.β == 0.0) and m.t == 0.0


@pytest.mark.parametrize("eta", [1e-3, 10.0, 1e4])
def test_regularisation_strengths(partls, oracle, eta):
    X, y, P, _ = oracle.synth(20260130, 400, 15, 3)
    _check(partls, oracle, X, y, P, eta=eta)


def test_badly_scaled_columns(partls, oracle):
    """columns differing by 1e6 in scale: the unit-diagonal scaling of the tableau absorbs it"""
    X, y, P, _ = oracle.synth(20260131, 500, 12, 3)
    X = np.asfortranarray(X * np.logspace(-3, 3, 12)[None, :])
    _check(partls, oracle, X, y, P)


def test_wide_tableau_boundary_sizes(partls, oracle):
    """n around the 16-wide tile boundaries of the register kernel (n = M or M+1)"""
    for M, K in [(15, 3), (16, 4), (17, 4), (31, 5), (32, 4), (33, 3)]:
        X, y, P, _ = oracle.synth(20260140 + M, 600, M, K)
        Xo, Po = oracle.homogeneous(X, P)
        R, z = oracle.compress(Xo, y)
        ref = oracle.opt_patterns(R, z, Po, np.arange(1 << (K + 1)))
        ctx = partls.default_context()
        ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
        bo, bp, allopt, unconv = ctx.opt_sweep(0, -1, want_all=True)
        assert unconv == 0
        np.testing.assert_allclose(allopt, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("noise,tol", [(1e-3, 1e-9), (1e-4, 1e-7)])
def test_ill_conditioned_model_parity(partls, oracle, noise, tol):
    """cond(Xo) ~ 1/noise (6e3 / 8e4): the Gram-based solve alone loses cond^2*eps digits in the model; the data-space
    refinement of the winner (refine_solution) restores the accuracy of the reference's QR-based NNLS."""
    rng = np.random.default_rng(42)
    N, D, K = 2000, 24, 4
    Z = rng.standard_normal((N, 6))
    X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))
    grp = np.arange(D) % K
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
    y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
    ref = oracle.fit_opt(X, y, P)
    for alg in (partls.Opt, partls.BnB):
        m, _, rep = partls.fit(alg, X, y, P)
        assert abs(rep.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        np.testing.assert_allclose(m.α, ref["alpha"], atol=tol)
        np.testing.assert_allclose(m.β, ref["beta"], atol=100 * tol)
        assert abs(m.t - ref["t"]) < tol


@pytest.mark.parametrize("seed", [42, 43, 44])
def test_beyond_the_gram_form_the_call_reports_it(partls, oracle, seed):
    """cond(Xo) from 7e5 to 7e7: the fp64 Gram form loses the smallest directions (cond^2 * eps -> 1) and with them columns the
    reference's QR-based NNLS (Opt.jl:89) would use.  The winner's KKT conditions are verified against the DATA: at each level either
    the fit equals the oracle's, or the call says PARTLS_ERR_ILL_CONDITIONED — never a silently different model."""
    reported = 0
    for noise in (1e-5, 3e-6, 1e-6, 3e-7, 1e-7):
        rng = np.random.default_rng(seed)
        N, D, K = 2000, 24, 4
        Z = rng.standard_normal((N, 6))
        X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))
        grp = np.arange(D) % K
        P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
        y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
        ref = oracle.fit_opt(X, y, P)
        for alg in (partls.Opt, partls.BnB):
            try:
                m, _, rep = partls.fit(alg, X, y, P, on_ill_conditioned="raise")
            except partls.PartlsError as e:
                assert e.status == partls.lowlevel.ERR_ILL_CONDITIONED, e
                assert partls.default_context().kkt_violation() > 1e-12
                reported += 1
                # the default: the same model comes back with a warning and the report says so (the reference returns a model here too)
                with pytest.warns(partls.IllConditionedWarning):
                    m2, _, rep2 = partls.fit(alg, X, y, P)
                assert rep2.ill_conditioned and rep2.kkt_violation > 1e-12 and np.isfinite(rep2.opt) and m2.α.shape == (D,)
                continue
            # a fit that passed the data-space check is never worse than the oracle's; it may be (slightly) BETTER: at cond > 1e6 the
            # oracle's own dependence rule (oracle/partls_oracle.h) drops columns a KKT-verified solution still uses
            assert rep.opt <= ref["opt"] * (1 + 1e-9), (noise, alg)
            if abs(rep.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]):
                np.testing.assert_allclose(m.α, ref["alpha"], atol=1e-6)
            assert partls.default_context().kkt_violation() <= 1e-12
    assert reported >= 4                                   # the two worst levels are certainly out of reach of the Gram form


@pytest.mark.parametrize("seed", [42, 43])
def test_alt_is_verified_against_the_data_too(partls, oracle, seed):
    """fit(Alt)'s alpha-step (tableau) and beta-step (K' x K' normal equations: condition SQUARED against the reference's QR solve,
    Alt.jl:110) both work on the Gram form; the last iteration is checked against the data (beta-step stationarity A'Xo'r = 0 and the
    alpha-step's KKT conditions).  From the same start as the oracle's dense Alt: at every conditioning level either the fit equals the
    oracle's or the call reports PARTLS_ERR_ILL_CONDITIONED — and well-conditioned data are never reported."""
    import warnings
    reported = 0
    for noise in (1e-2, 1e-4, 1e-6, 1e-7):
        rng = np.random.default_rng(seed)
        N, D, K = 2000, 24, 4
        Z = rng.standard_normal((N, 6))
        X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))
        grp = np.arange(D) % K
        P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
        y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
        r2 = np.random.default_rng(seed + 1000)
        a0, b0 = r2.random(D + 1), (r2.random(K + 1) - 0.5) * 10
        ref = oracle.fit_alt(X, y, P, a0, b0, eps=1e-9, T=60)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", partls.IllConditionedWarning)
            m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0, eps=1e-9, T=60)
        kkt = partls.default_context().kkt_violation()
        if rep.get("ill_conditioned"):
            assert kkt > 1e-12 and noise <= 1e-4, (noise, kkt)
            reported += 1
            with pytest.raises(partls.PartlsError) as ei:
                partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0, eps=1e-9, T=60, on_ill_conditioned="raise")
            assert ei.value.status == partls.lowlevel.ERR_ILL_CONDITIONED
            continue
        assert kkt <= 1e-12
        # unreported: the iterate is a verified fixed-point step; Alt is a local method, so the comparison with the oracle's run is on
        # the loss (both runs descend from the same start; beyond cond ~ 1e5 their paths may part without either being wrong)
        if noise >= 1e-4:
            assert abs(rep.opt - ref["opt"]) <= 1e-7 * max(1.0, ref["opt"]), (noise, rep.opt, ref["opt"])
    assert reported >= 1


@pytest.mark.parametrize("M", [1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300, 385])
def test_gram_virtual_columns_at_tile_boundaries(partls, M):
    """The ones / y columns of Z = [X 1 y] are virtual (they ride on the diagonal 128 x 128 tiles as extra 16 x 16 accumulators); the
    feature tiles are 128 wide with a zero-padded edge tile, the diagonal tiles store their upper triangle at 16 x 16 granularity.
    M around every multiple of 16 / 64 / 128 that matters (round 1: at M % 64 == 63 G[M][M] came out 0, found by the fuzz
    campaign).  N = 3 M + 17 is never a multiple of the 16-sample panel: the last panel of the data is partial.  Gram block
    against numpy."""
    rng = np.random.default_rng(700 + M)
    N, K = 3 * M + 17, 5
    X = rng.standard_normal((N, M))
    y = X @ rng.standard_normal(M) + 2.0 + 0.1 * rng.standard_normal(N)
    P = np.zeros((M, K), dtype=np.int64)
    P[np.arange(M), np.arange(M) % K] = 1
    ctx = partls.Context()
    ctx.opt_prepare(X, y, P, 0.0, 0)
    G = ctx.gram()
    Z = np.column_stack([X, np.ones(N), y])
    Gn = Z.T @ Z
    d = np.sqrt(np.outer(np.diag(Gn), np.diag(Gn)))
    assert np.max(np.abs(G - Gn) / d) < 1e-13


def test_strided_inputs_through_the_c_abi(partls):
    """ldX > N and ldP > M (sub-matrices of larger Julia arrays): host pointers are compacted by a 2-D copy, device pointers are
    used in place with their stride by the Gram / residual / X'r kernels.  Results must equal the contiguous call."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(77)
    N, M, K, padN, padM = 333, 37, 4, 9, 3
    X = rng.standard_normal((N, M))
    P = np.zeros((M, K), dtype=np.int64); P[np.arange(M), np.arange(M) % K] = 1
    y = X @ (rng.random(M) * np.array([1.0, -2.0, 0.5, -1.0])[np.arange(M) % K]) + 0.4 + 0.05 * rng.standard_normal(N)
    ref_model, _, ref = partls.fit(partls.Opt, X, y, P)
    Xbig = np.asfortranarray(rng.standard_normal((N + padN, M)))
    Xbig[:N, :] = X
    Pbig = np.asfortranarray(np.full((M + padM, K), 7, dtype=np.int64))
    Pbig[:M, :] = P
    lib = partls.lowlevel.lib()
    h = C.c_void_p()
    assert lib.partls_create(0, C.byref(h)) == 0
    try:
        for on_device in (0, 1):
            if on_device:
                dX = torch.from_numpy(np.ascontiguousarray(Xbig.T)).to("cuda:0")      # memory image == column-major Xbig
                dy = torch.from_numpy(y.copy()).to("cuda:0")
                xp, yp = C.c_void_p(dX.data_ptr()), C.c_void_p(dy.data_ptr())
            else:
                xp, yp = C.c_void_p(Xbig.ctypes.data), C.c_void_p(y.ctypes.data)
            st = lib.partls_opt_prepare(h, xp, N, M, N + padN, yp, on_device, C.c_void_p(Pbig.ctypes.data), K, M + padM,
                                        C.c_double(0.0), 0)
            assert st == 0, lib.partls_last_error()
            bo, bp, nu = C.c_double(), C.c_int64(), C.c_int64()
            assert lib.partls_opt_sweep(h, 0, -1, C.byref(bo), C.byref(bp), None, C.byref(nu)) == 0
            a = np.zeros(M); b = np.zeros(K); t = C.c_double(); o = C.c_double(); bi = C.c_int64()
            assert lib.partls_opt_finish(h, bp.value, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                         C.byref(t), C.byref(o), C.byref(bi)) == 0
            assert abs(o.value - ref.opt) <= 1e-12 * max(1.0, ref.opt)
            np.testing.assert_allclose(a, ref_model.α, atol=1e-10)
            np.testing.assert_allclose(b, ref_model.β, atol=1e-9)
            assert abs(t.value - ref_model.t) < 1e-10
    finally:
        lib.partls_destroy(h)


def test_refinement_by_tableau_inverse_equals_refinement_by_cholesky(partls, oracle):
    """The winner's data-space refinement solves its correction equations with the inverse the node solve left in its final tableau
    (register kernel, or the cooperative kernel's global image beyond n = 320); PARTLS_NO_TAB_REFINE (read at context creation) forces
    the older host Cholesky of G_BB.  Both must land on the
    same model — on a well-conditioned problem and on one with cond(X) ~ 1e4, where the un-refined Gram solution is only good to 1e-8
    — in both intercept modes, and agree with the dense oracle."""
    import os
    rng = np.random.default_rng(4242)
    for cond, N, D, K in ((1.0, 400, 24, 4), (1e4, 400, 24, 4), (1e3, 700, 330, 3)):   # the last: n > 320, cooperative kernel's tableau
        U, _ = np.linalg.qr(rng.standard_normal((N, D)))
        V, _ = np.linalg.qr(rng.standard_normal((D, D)))
        X = (U * np.geomspace(1.0, 1.0 / cond, D)) @ V.T * np.sqrt(N)
        P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
        y = X @ (rng.random(D) * np.array([1.0, -2.0, 0.5, 3.0])[np.arange(D) % K]) + 1.5 + 0.05 * rng.standard_normal(N)   # (K <= 4)
        ref = oracle.fit_opt(X, y, P)
        got = {}
        for mode in ("tab", "chol"):
            if mode == "chol":
                os.environ["PARTLS_NO_TAB_REFINE"] = "1"
            try:
                ctx = partls.Context()
            finally:
                os.environ.pop("PARTLS_NO_TAB_REFINE", None)
            for flags in (0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT):
                ctx.opt_prepare(X, y, P, 0.0, flags)
                bo, bp, _, unconv = ctx.opt_sweep(0, -1)
                assert unconv == 0
                got[(mode, flags)] = ctx.opt_finish(bp)
            ctx.close()
        for flags in (0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT):
            a1, b1, t1, o1, i1 = got[("tab", flags)]
            a2, b2, t2, o2, i2 = got[("chol", flags)]
            assert i1 == i2
            assert abs(o1 - o2) <= 1e-12 * max(1.0, o2)
            np.testing.assert_allclose(a1, a2, atol=1e-9)
            np.testing.assert_allclose(b1, b2, atol=1e-9 * max(1.0, np.abs(b2).max()))
            assert abs(t1 - t2) <= 1e-9 * max(1.0, abs(t2))
            assert abs(o1 - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


def test_alt_certificate_or_data_check(partls, oracle, monkeypatch):
    """fit(Alt) on a Gaussian design: Gershgorin's bound on the scaled Gram block (radius < 0.75: cond <= 7) certifies the Gram-form solves
    and the extra pass over X is skipped (kkt_violation reads exactly 0); with PARTLS_ALT_ALWAYS_CHECK the last iteration is verified
    against the data instead (a tiny, non-zero violation).  Same model either way, equal to the oracle's dense Alt from the same start."""
    rng = np.random.default_rng(11)
    N, D, K = 20000, 30, 5
    X = rng.standard_normal((N, D))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    y = X @ (rng.random(D) * (np.arange(D) % K - 2.0)) + 0.3 * rng.standard_normal(N)
    a0, b0 = rng.random(D + 1), (rng.random(K + 1) - 0.5) * 10
    ref = oracle.fit_alt(X, y, P, a0, b0, eps=1e-9, T=50)
    res = {}
    for mode in ("certified", "checked"):
        if mode == "checked":
            monkeypatch.setenv("PARTLS_ALT_ALWAYS_CHECK", "1")
        ctx = partls.Context(0)
        ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
        res[mode] = ctx.alt_prepared(a0, b0, eps=1e-9, T=50) + (ctx.kkt_violation(),)
        ctx.close()
    assert res["certified"][5] == 0.0 and 0.0 < res["checked"][5] <= 1e-13
    for mode in res:
        a, b, t, opt, it, _ = res[mode]
        assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        np.testing.assert_allclose(a, ref["alpha"], atol=1e-7)
    np.testing.assert_array_equal(res["certified"][0], res["checked"][0])


@pytest.mark.parametrize("D,K,kind", [(170, 5, "plain"), (200, 4, "dup"), (240, 5, "null"), (257, 4, "triple"), (270, 5, "scaled"), (271, 3, "dup"), (288, 4, "plain")])
def test_winner_from_the_512_thread_kernel_mid_sizes(partls, oracle, monkeypatch, D, K, kind):
    """n = 171 .. 289 tableau variables run on the 512-thread register kernel, which since round 4 leaves the solution of its winner behind
    (T <= 17 tile columns; 288 features: T = 18, the re-solve path): the finish starts from it — accepted only with the winning pattern's
    signs, refined / verified in data space like a fresh solve.  Against the oracle and against the same fit with PARTLS_NO_EXPORT, with
    duplicate, null, dependent and badly scaled columns and eta > 0."""
    rng = np.random.default_rng(1000 + D)
    N = 2 * D + 50
    X = rng.standard_normal((N, D))
    if kind == "dup":
        X[:, 7] = X[:, 3]; X[:, D - 1] = X[:, D - 2]
    elif kind == "null":
        X[:, 5] = 0.0
    elif kind == "triple":
        X[:, 11] = 0.5 * X[:, 12] - 2.0 * X[:, 13]
    elif kind == "scaled":
        X *= np.exp(rng.uniform(-3, 3, size=D))[None, :]
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), rng.integers(0, K, size=D)] = 1
    y = X @ (rng.random(D) * (rng.random(K) - 0.5)[P.argmax(1)] * 6) + 0.7 + 0.5 * rng.standard_normal(N)
    eta = 0.3 if kind == "plain" else 0.0
    ref = oracle.fit_opt(X, y, P, eta=eta)
    res = {}
    for mode in ("export", "resolve"):
        if mode == "resolve":
            monkeypatch.setenv("PARTLS_NO_EXPORT", "1")
        ctx = partls.Context(0)
        try:
            ctx.opt_prepare(X, y, P, eta, 0)
            bo, bp, _, unconv = ctx.opt_sweep(0, -1)
            assert unconv == 0
            res[mode] = ctx.opt_finish(bp)
        finally:
            ctx.close()
    for mode, (a, b, t, opt, bi) in res.items():
        assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]), mode
        np.testing.assert_allclose(partls.predict(partls.PartLSFitResult(a, b, t, P), X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]),
                                   atol=1e-6 * np.linalg.norm(y), err_msg=mode)
    assert res["export"][4] == res["resolve"][4] and abs(res["export"][3] - res["resolve"][3]) <= 1e-12 * res["resolve"][3]
