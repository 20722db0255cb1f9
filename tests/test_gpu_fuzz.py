"""Randomised differential parity (GPU): the HIP path against the CPU oracle on many small random problems — random shapes
(1..17 tile columns), unequal group sizes, eta values, noise levels, dependent / duplicate / null columns.  Every pattern's
objective (faithful intercept: the reference's 2^(K+1) enumeration, Opt.jl:85-96) within 1e-8 relative (+2e-7 ||y|| absolute
at zero objectives, where the Gram form has abs error ~ sqrt(eps * yy)), same winner up to ties, model within 1e-6.
Seeds are fixed: failures reproduce (tools/fuzz_triage.py arbitrates a mismatch with KKT certificates; tools/tableau_emul.py
replays the device's pivoting decisions in numpy).  The whole campaign — 160 blocks, 2720 problems — runs by default
(PARTLS_FUZZ_BLOCKS narrows it).  What it found: the Gram kernel's virtual ones column at M % 64 == 63, the oracle's classic
independence test on exactly dependent columns, and (round 1, blocks 9 and 24) an exactly dependent column next to a nearly
collinear pair, whose computed pivot lands just above the fixed 1e-11 threshold: fixed in round 2 by the leave-one-out
acceptance rule of the tableau kernels (sweep_blk.hip header) — both problems are named regression tests below."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_problem(rng):
    K = int(rng.integers(1, 8))
    sizes = rng.integers(1, 14, size=K)
    if rng.random() < 0.3:
        sizes[rng.integers(0, K)] = int(rng.integers(14, 40))          # one big group (disables the two-level split or not)
    D = int(sizes.sum())
    N = int(rng.integers(D + 5, 6 * D + 40))
    P = np.zeros((D, K), dtype=np.int64)
    order = rng.permutation(D) if rng.random() < 0.5 else np.arange(D)  # groups need not be contiguous in the data
    c = 0
    for k, s in enumerate(sizes):
        P[order[c:c + s], k] = 1
        c += s
    X = rng.standard_normal((N, D))
    if rng.random() < 0.4:
        X *= np.exp(rng.uniform(-3, 3, size=D))[None, :]                # badly scaled columns
    kind = rng.random()
    if kind < 0.15 and D >= 3:
        X[:, rng.integers(0, D)] = X[:, rng.integers(0, D)]             # duplicate column
    elif kind < 0.3 and D >= 3:
        X[:, rng.integers(0, D)] = 0.0                                  # null column
    elif kind < 0.4 and D >= 4:
        i, j, l = rng.choice(D, 3, replace=False)
        X[:, i] = 0.5 * X[:, j] - 2.0 * X[:, l]                         # dependent triple
    grp = P.argmax(1)
    a = rng.random(D)
    beta = (rng.random(K) - 0.5) * 10
    y = X @ (a * beta[grp]) + rng.uniform(-2, 2) + rng.choice([0.0, 1e-3, 0.3, 3.0]) * rng.standard_normal(N)
    eta = float(rng.choice([0.0, 0.0, 1e-3, 0.5]))
    return X, y, P, eta


@pytest.mark.parametrize("block", range(int(os.environ.get("PARTLS_FUZZ_BLOCKS", "160"))))
def test_fuzz_opt_all_patterns_vs_oracle(partls, oracle, block):
    rng = np.random.default_rng(9000 + block)
    for it in range(12):
        X, y, P, eta = _random_problem(rng)
        ref = oracle.fit_opt(X, y, P, eta=eta, return_all=True)
        scale = max(1.0, float(np.linalg.norm(y)))
        model, _, rep = partls.fit(partls.Opt, X, y, P, η=eta, returnAllSolutions=True)
        got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
        tag = f"block {block} it {it} shape {X.shape} K {P.shape[1]} eta {eta}"
        rtol = 1e-8
        np.testing.assert_allclose(got, ref["all_opt"], rtol=rtol, atol=2e-7 * scale, err_msg=tag)
        assert abs(got.min() - ref["opt"]) <= rtol * max(1.0, ref["opt"]) + 2e-7 * scale, tag
        # the reported winner attains the minimum (ties may pick another index of equal objective)
        assert ref["all_opt"][int(np.argmin(got))] <= ref["opt"] + rtol * max(1.0, ref["opt"]) + 2e-7 * scale, tag


@pytest.mark.parametrize("block", range(int(os.environ.get("PARTLS_FUZZ_BLOCKS", "160")) // 2))
def test_fuzz_fit_model_vs_oracle(partls, oracle, block):
    """default (free-intercept) fit: optimum and predictions against the oracle's dense Lawson–Hanson path."""
    rng = np.random.default_rng(9500 + block)
    for it in range(10):
        X, y, P, eta = _random_problem(rng)
        ref = oracle.fit_opt(X, y, P, eta=eta)
        model, _, rep = partls.fit(partls.Opt, X, y, P, η=eta)
        tag = f"block {block} it {it} shape {X.shape} K {P.shape[1]} eta {eta}"
        scale = max(1.0, float(np.linalg.norm(y)))
        assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"]) + 1e-9 * scale, tag
        yh = partls.predict(model, X)
        yr = oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"])
        # predictions agree wherever the optimum is unique; with dependent columns only the fit is unique, so compare fits
        assert np.linalg.norm(yh - yr) <= 1e-6 * scale, tag


@pytest.mark.parametrize("block,it,shape", [(9, 1, (111, 23)), (24, 5, (37, 26))])
def test_regression_dependent_column_next_to_collinear_pair(partls, oracle, block, it, shape):
    """Round-1 failures (3 % and 10 % off on 2 of 16 / 13 of 64 patterns): a badly scaled dependent triple x_i = 0.5 x_j - 2 x_l
    makes x_i and x_l nearly collinear on the unit-diagonal scale (pivot ~1.7e-5) and x_j exactly dependent on the pair with
    regression coefficients ~220, so its Gram-form pivot carries an error of eps * 220^2 and comes out at 2..5e-11 — above the
    fixed 1e-11 rejection.  The leave-one-out rule (d_k > 1e-11 * max(1, c_j^2)) refuses it, as the reference's QR-based NNLS
    does (Opt.jl:89).  Checked for every pattern, in both intercept modes, on both tableau kernels."""
    rng = np.random.default_rng(9000 + block)
    for _ in range(it + 1):
        X, y, P, eta = _random_problem(rng)
    assert X.shape == shape and np.linalg.matrix_rank(X) == X.shape[1] - 1
    ref = oracle.fit_opt(X, y, P, eta=eta, return_all=True)
    scale = max(1.0, float(np.linalg.norm(y)))
    for generic in (False, True):
        model, _, rep = partls.fit(partls.Opt, X, y, P, η=eta, returnAllSolutions=True, generic_kernel=generic)
        got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
        np.testing.assert_allclose(got, ref["all_opt"], rtol=1e-8, atol=2e-7 * scale, err_msg=f"generic={generic}")
        m2, _, r2 = partls.fit(partls.Opt, X, y, P, η=eta, generic_kernel=generic)
        assert abs(r2.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"]) + 1e-9 * scale
        assert np.linalg.norm(partls.predict(m2, X) - oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"])) <= 1e-6 * scale


def _wellposed_problem(rng, dmax=40):
    """full-rank random problem (no planted dependencies) with unequal groups and odd N — for the paths whose iterates, not
    only whose optimum, are compared"""
    K = int(rng.integers(1, 7))
    sizes = rng.integers(1, max(2, dmax // K), size=K)
    D = int(sizes.sum())
    N = int(rng.integers(D + 7, 5 * D + 61))
    P = np.zeros((D, K), dtype=np.int64)
    P[rng.permutation(D), np.repeat(np.arange(K), sizes)] = 1
    X = rng.standard_normal((N, D)) * np.exp(rng.uniform(-1.5, 1.5, size=D))[None, :]
    grp = P.argmax(1)
    y = X @ (rng.random(D) * ((rng.random(K) - 0.5) * 6)[grp]) + rng.uniform(-1, 1) + 0.2 * rng.standard_normal(N)
    return X, y, P


@pytest.mark.parametrize("block", range(max(3, int(os.environ.get("PARTLS_FUZZ_BLOCKS", "160")) // 2)))
def test_fuzz_bnb_equals_opt_and_oracle(partls, oracle, block):
    """BnB.jl:30-132 returns the optimum of the same problem as Opt (with the signed intercept of BnB.jl:36-39)."""
    rng = np.random.default_rng(9700 + block)
    for it in range(8):
        X, y, P = _wellposed_problem(rng)
        eta = float(rng.choice([0.0, 0.1]))
        ref = oracle.fit_bnb(X, y, P, eta=eta)
        m1, _, r1 = partls.fit(partls.BnB, X, y, P, η=eta)
        m2, _, r2 = partls.fit(partls.Opt, X, y, P, η=eta)
        tag = f"block {block} it {it} shape {X.shape} K {P.shape[1]} eta {eta}"
        assert abs(r1.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"]), tag
        assert abs(r1.opt - r2.opt) <= 1e-8 * max(1.0, r2.opt), tag
        np.testing.assert_allclose(partls.predict(m1, X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]),
                                   atol=1e-6 * max(1.0, float(np.linalg.norm(y))), err_msg=tag)


@pytest.mark.parametrize("block", range(max(3, int(os.environ.get("PARTLS_FUZZ_BLOCKS", "160")) // 2)))
def test_fuzz_alt_same_start_vs_oracle(partls, oracle, block):
    """Alt.jl:50-124 from the same (alpha0, beta0): the alternating NNLS / least-squares iterates are deterministic, so the
    final objective and model agree with the oracle's."""
    rng = np.random.default_rng(9800 + block)
    for it in range(8):
        X, y, P = _wellposed_problem(rng, dmax=30)
        M, K = P.shape
        a0 = rng.random(M + 1)
        b0 = (rng.random(K + 1) - 0.5) * 10
        T = int(rng.integers(1, 12))
        ref = oracle.fit_alt(X, y, P, a0, b0, eta=0.0, eps=1e-9, T=T)
        m, _, rep = partls.fit(partls.Alt, X, y, P, η=0.0, ϵ=1e-9, T=T, alpha0=a0, beta0=b0)
        tag = f"block {block} it {it} shape {X.shape} K {K} T {T}"
        assert abs(rep.opt - ref["opt"]) <= 1e-7 * max(1.0, ref["opt"]), tag
        np.testing.assert_allclose(partls.predict(m, X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]),
                                   atol=1e-6 * max(1.0, float(np.linalg.norm(y))), err_msg=tag)


@pytest.mark.parametrize("D", [273, 300, 305, 340])
def test_fuzz_large_n_generic_path_vs_oracle(partls, oracle, D):
    """the top of the register kernel's range (n = 274, 301, 306: T = 18, 19, 20 tile columns) and, beyond n = 320, the global-memory
    tableau kernels (sweep_generic.hip chains, sweep_coop.hip single nodes)."""
    rng = np.random.default_rng(9900 + D)
    K = 3
    N = 2 * D + 11
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), rng.integers(0, K, size=D)] = 1
    X = rng.standard_normal((N, D))
    grp = P.argmax(1)
    y = X @ (rng.random(D) * np.array([2.0, -1.0, 0.5])[grp]) + 0.7 + 0.1 * rng.standard_normal(N)
    ref = oracle.fit_opt(X, y, P, return_all=True)
    model, _, rep = partls.fit(partls.Opt, X, y, P, returnAllSolutions=True)
    got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
    np.testing.assert_allclose(got, ref["all_opt"], rtol=1e-8, atol=1e-8)
    m2, _, r2 = partls.fit(partls.Opt, X, y, P)
    assert abs(r2.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


@pytest.mark.parametrize("D", [300, 305, 319])
def test_top_of_the_register_kernel_behind_its_knob(partls, oracle, D):
    """Since round 4 the library leaves the register kernel at T = 18 tile columns (n <= 288): its T = 19 / 20 instantiations spill and lose
    to the deferred-update kernel.  They stay compiled behind PARTLS_REG_MAXT (read at context creation): the same problem on both
    kernels, every pattern against the oracle, and the two kernels against each other."""
    import os
    rng = np.random.default_rng(9900 + D)
    K = 3
    N = 2 * D + 11
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), rng.integers(0, K, size=D)] = 1
    X = rng.standard_normal((N, D))
    y = X @ (rng.random(D) * np.array([2.0, -1.0, 0.5])[P.argmax(1)]) + 0.7 + 0.1 * rng.standard_normal(N)
    ref = oracle.fit_opt(X, y, P, return_all=True)
    got = {}
    for maxt in ("18", "20"):
        os.environ["PARTLS_REG_MAXT"] = maxt
        try:
            ctx = partls.Context()
        finally:
            os.environ.pop("PARTLS_REG_MAXT", None)
        ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
        bo, bp, all_opt, unconv = ctx.opt_sweep(0, -1, want_all=True)
        assert unconv == 0
        got[maxt] = (all_opt.copy(), ctx.opt_finish(bp))
        ctx.close()
    for maxt in ("18", "20"):
        np.testing.assert_allclose(got[maxt][0], ref["all_opt"], rtol=1e-8, atol=1e-8)
        assert abs(got[maxt][1][3] - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
    np.testing.assert_allclose(got["18"][0], got["20"][0], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(got["18"][1][0], got["20"][1][0], atol=1e-8)
