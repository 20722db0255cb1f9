"""CPU tests: the oracle (oracle/partls_oracle.c) against the reference's known answers and the scipy-made golden fixtures."""
import numpy as np
import pytest

from conftest import load_golden

SYNTH = ["synth_a", "synth_b", "synth_eta", "synth_c", "corr"]


def test_toy_known_answers_from_reference_tests(oracle):
    """test/runtests.jl:9-36 — fit(Opt, X, y, P, η=0.0): opt ≈ 0 (atol 1e-6) and sum(ŷ - y)^2 ≈ 0 (atol 1e-6)."""
    g = load_golden("toy")
    r = oracle.fit_opt(g["X"], g["y"], g["P"], eta=0.0, return_all=True)
    assert abs(r["opt"]) < 1e-6
    yhat = oracle.predict(g["X"], g["P"], r["alpha"], r["beta"], r["t"])
    assert abs(np.sum(yhat - g["y"]) ** 2) < 1e-6
    # exact rationals (Xo is 4x4 nonsingular: w = Xo^-1 y), SURVEY.md §8c
    np.testing.assert_allclose(r["alpha"], g["exact_alpha"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["beta"], g["exact_beta"], rtol=0, atol=1e-12)
    assert abs(r["t"] - g["exact_t"]) < 1e-12
    assert r["best_index"] == 5
    np.testing.assert_allclose(r["all_opt"], g["opt_all_opt"], rtol=0, atol=1e-12)


def test_toy_alt_and_bnb(oracle):
    """runtests.jl:136-144 — Alt and BnB reach opt ≈ 0 on the toy (Alt with the explicit init of SURVEY.md §8a)."""
    g = load_golden("toy")
    ra = oracle.fit_alt(g["X"], g["y"], g["P"], g["alt_alpha0"], g["alt_beta0"])
    assert abs(ra["opt"]) < 1e-6
    np.testing.assert_allclose(ra["alpha"], g["exact_alpha"], atol=1e-9)
    np.testing.assert_allclose(ra["beta"], g["exact_beta"], atol=1e-9)
    ra1 = oracle.fit_alt(g["X"], g["y"], g["P"], g["alt_alpha0"], g["alt1_beta0"], T=1)
    assert abs(ra1["opt"] - 0.08609555601316027) < 1e-12          # SURVEY.md §8a probe value
    assert abs(ra1["opt"] - float(g["alt1_opt"])) < 1e-12
    rb = oracle.fit_bnb(g["X"], g["y"], g["P"])
    assert abs(rb["opt"]) < 1e-6 and rb["nopen"] == 1
    np.testing.assert_allclose(rb["alpha"], g["exact_alpha"], atol=1e-12)
    np.testing.assert_allclose(rb["beta"], g["exact_beta"], atol=1e-12)


def test_toy_return_all_solutions(oracle):
    g = load_golden("toy")
    r = oracle.fit_opt(g["X"], g["y"], g["P"], all_models=True)
    np.testing.assert_allclose(r["all_alpha"], g["opt_all_alpha"], atol=1e-11)
    np.testing.assert_allclose(r["all_beta"], g["opt_all_beta"], atol=1e-11)
    np.testing.assert_allclose(r["all_t"], g["opt_all_t"], atol=1e-11)


@pytest.mark.parametrize("name", SYNTH)
def test_opt_against_scipy_golden(oracle, name):
    g = load_golden(name)
    eta = float(g.get("eta", 0.0))
    r = oracle.fit_opt(g["X"], g["y"], g["P"], eta=eta, return_all=True)
    scale = max(1.0, float(g["opt_opt"]))
    np.testing.assert_allclose(r["all_opt"], g["opt_all_opt"], rtol=0, atol=1e-9 * scale)
    assert r["best_index"] == int(g["opt_best_index"])
    np.testing.assert_allclose(r["alpha"], g["opt_alpha"], atol=1e-8)
    np.testing.assert_allclose(r["beta"], g["opt_beta"], atol=1e-8)
    assert abs(r["t"] - float(g["opt_t"])) < 1e-8


@pytest.mark.parametrize("name", ["synth_a", "synth_b", "synth_eta", "synth_c"])
def test_alt_against_scipy_golden(oracle, name):
    g = load_golden(name)
    r = oracle.fit_alt(g["X"], g["y"], g["P"], g["alt_alpha0"], g["alt_beta0"], eta=float(g["eta"]))
    assert abs(r["opt"] - float(g["alt_opt"])) < 1e-8 * max(1.0, float(g["alt_opt"]))
    assert r["iters"] == int(g["alt_iters"])
    np.testing.assert_allclose(r["alpha"], g["alt_alpha"], atol=1e-7)
    np.testing.assert_allclose(r["beta"], g["alt_beta"], atol=1e-7)


@pytest.mark.parametrize("name", SYNTH)
def test_bnb_against_scipy_golden(oracle, name):
    g = load_golden(name)
    r = oracle.fit_bnb(g["X"], g["y"], g["P"], eta=float(g.get("eta", 0.0)))
    assert abs(r["opt"] - float(g["bnb_opt"])) < 1e-9 * max(1.0, float(g["bnb_opt"]))
    assert r["nopen"] == int(g["bnb_nopen"])
    np.testing.assert_allclose(r["alpha"], g["bnb_alpha"], atol=1e-7)
    np.testing.assert_allclose(r["beta"], g["bnb_beta"], atol=1e-7)
    # identity: the BnB optimum equals the Opt optimum
    assert abs(r["opt"] - float(g["opt_opt"])) < 1e-8 * max(1.0, float(g["opt_opt"]))


def test_nnls_kkt_certificate(oracle):
    """Solver-independent certificate: x >= 0, w = A'(b - Ax) <= tol, x∘w ≈ 0."""
    rng = np.random.default_rng(3)
    for m, n in [(40, 12), (120, 60), (25, 25), (15, 30)]:
        A = rng.standard_normal((m, n)); b = rng.standard_normal(m)
        x, rn, mode, _ = oracle.nnls(A, b)
        assert mode == 0 and np.all(x >= 0)
        w = A.T @ (b - A @ x)
        assert w.max() < 1e-10
        assert np.abs(x * w).max() < 1e-10
        assert abs(rn - np.linalg.norm(A @ x - b)) < 1e-10


def test_compressed_equals_dense(oracle):
    """||Xo w - y|| == ||R w - z||: the QR-compressed problem gives the same per-pattern objectives."""
    X, y, P, _ = oracle.synth(20260105, 800, 18, 4)
    Xo, Po = oracle.homogeneous(X, P)
    dense = oracle.opt_patterns(Xo, y, Po, np.arange(32))
    R, z = oracle.compress(Xo, y)
    comp = oracle.opt_patterns(R, z, Po, np.arange(32))
    np.testing.assert_allclose(comp, dense, rtol=1e-11)


def test_free_intercept_identity(oracle):
    """min over the ± intercept pair == optimum with the intercept left free: checked through BnB-style
    relaxation, i.e. min(all_opt[b], all_opt[b + 2^K]) is what a 2^K enumeration with a free intercept returns."""
    g = load_golden("synth_b")
    K = g["P"].shape[1]
    ao = g["opt_all_opt"]
    pair_min = np.minimum(ao[: 1 << K], ao[1 << K:])
    assert abs(pair_min.min() - float(g["opt_opt"])) < 1e-12


def test_synth_generator_properties(oracle):
    X, y, P, ws = oracle.synth(20260002, 4000, 16, 4)
    assert abs(X.mean()) < 0.02 and abs(X.std() - 1.0) < 0.02
    assert P.sum(axis=1).tolist() == [1] * 16 and P.sum(axis=0).tolist() == [4, 4, 4, 4]
    resid = y - X @ ws - 1.0
    assert abs(resid.std() - 0.1) < 0.01
    X2, y2, _, _ = oracle.synth(20260002, 4000, 16, 4)
    assert np.array_equal(X, X2) and np.array_equal(y, y2)


def _fuzz_problem(rng):
    """Small random problem with unequal groups, badly scaled and (sometimes) duplicate / null / dependent columns — the same
    family tests/test_gpu_fuzz.py throws at the HIP path."""
    K = int(rng.integers(1, 6))
    sizes = rng.integers(1, 9, size=K)
    D = int(sizes.sum())
    N = int(rng.integers(D + 5, 4 * D + 30))
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), np.repeat(np.arange(K), sizes)] = 1
    X = rng.standard_normal((N, D)) * np.exp(rng.uniform(-2, 2, size=D))[None, :]
    kind = rng.random()
    if kind < 0.3 and D >= 3:
        X[:, rng.integers(0, D)] = X[:, rng.integers(0, D)]
    elif kind < 0.5 and D >= 3:
        X[:, rng.integers(0, D)] = 0.0
    elif kind < 0.8 and D >= 4:
        i, j, l = rng.choice(D, 3, replace=False)
        X[:, i] = 0.5 * X[:, j] - 2.0 * X[:, l]
    y = X @ rng.standard_normal(D) + 1.0 + 0.3 * rng.standard_normal(N)
    return X, y, P


def test_nnls_kkt_certificate_on_rank_deficient_problems(oracle):
    """The oracle is only a valid checker if its NNLS returns the optimum.  KKT is necessary and sufficient for this convex
    problem, so every pattern of 40 random (mostly rank-deficient) problems is certified without any other solver: x >= 0,
    A'(b - Ax) <= 0 on the zero set, = 0 on the support, and the reported residual is the true one.  (The classic
    Lawson–Hanson independence test alone fails this on exactly dependent columns — found by the GPU fuzz test — hence the
    additional rejection rule documented in partls_oracle.c.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzzgen", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    problems = []
    for seed, it in ((9015, 4), (9069, 1), (9024, 5), (9009, 0)):       # the problems on which the classic test failed
        r = np.random.default_rng(seed)
        for _ in range(it + 1):
            Xf, yf, Pf, eta = fz._random_problem(r)
        if eta == 0.0:
            problems.append((Xf, yf, Pf))
    rng = np.random.default_rng(4242)
    problems += [_fuzz_problem(rng) for _ in range(30)]
    for X, y, P in problems:
        Xo, Po = oracle.homogeneous(X, P)
        K1 = Po.shape[1]
        ref = oracle.fit_opt(X, y, P, return_all=True)
        for b in range(1 << K1):
            s = np.array([1.0 if (b >> k) & 1 else -1.0 for k in range(K1)])
            A = Xo * (Po @ s)[None, :]
            x, rn, mode, _ = oracle.nnls(A, y)
            r = y - A @ x
            g = A.T @ r
            tol = 1e-9 * (np.linalg.norm(A, axis=0) + 1e-300) * max(1.0, np.linalg.norm(y))
            assert mode == 0 and x.min() >= 0.0
            assert np.all(g[x == 0] <= tol[x == 0]) and np.all(np.abs(g[x > 0]) <= tol[x > 0])
            assert abs(np.linalg.norm(r) - rn) <= 1e-9 * max(1.0, rn)
            assert abs(ref["all_opt"][b] - rn) <= 1e-9 * max(1.0, rn)
