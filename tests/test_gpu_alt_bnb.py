"""GPU parity tests for the §8f rows: fit(Alt) and fit(BnB) on the Gram/tableau kernels, against the reference's toy known
answers (test/runtests.jl:41-69,123-146), the scipy-made golden fixtures and the CPU oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
SYNTH = ["synth_a", "synth_b", "synth_eta", "synth_c"]


def test_toy_alt_known_answer(partls):
    """runtests.jl:41-69 for Alt: opt ≈ 0 (atol 1e-6), sum(ŷ - y)^2 ≈ 0; init given explicitly."""
    g = load_golden("toy")
    result = partls.fit(partls.Alt, g["X"], g["y"], g["P"], η=0.0, ϵ=1e-6, T=100, alpha0=g["alt_alpha0"], beta0=g["alt_beta0"])
    assert abs(result[2].opt) < 1e-6
    y_pred = partls.predict(result[0], g["X"])
    assert abs(np.sum(y_pred - g["y"]) ** 2) < 1e-6
    np.testing.assert_allclose(result[0].α, g["exact_alpha"], atol=1e-8)
    np.testing.assert_allclose(result[0].β, g["exact_beta"], atol=1e-8)
    assert abs(result[0].t - float(g["exact_t"])) < 1e-8


def test_toy_alt_first_iteration_loss(partls):
    g = load_golden("toy")
    result = partls.fit(partls.Alt, g["X"], g["y"], g["P"], T=1, alpha0=g["alt_alpha0"], beta0=g["alt1_beta0"])
    assert abs(result[2].opt - 0.08609555601316027) < 1e-10
    np.testing.assert_allclose(result[0].α, g["alt1_alpha"], atol=1e-9)
    np.testing.assert_allclose(result[0].β, g["alt1_beta"], atol=1e-9)


def test_toy_alt_float32_and_seeded_rng(partls):
    """runtests.jl:123-146: Float32 inputs; rng=Int seeds the start (Alt.jl:58-66). Alt is a local method, so only the
    reproducibility of a seeded run is asserted here (runtests.jl:102-121)."""
    g = load_golden("toy")
    X32, y32 = g["X"].astype(np.float32), g["y"].astype(np.float32)
    r1 = partls.fit(partls.Alt, X32, y32, g["P"], rng=123, ϵ=1e-3, T=100)
    r2 = partls.fit(partls.Alt, X32, y32, g["P"], rng=123, ϵ=1e-3, T=100)
    assert abs(r1[2].opt - r2[2].opt) < 1e-6
    assert np.all(np.isfinite(r1[0].α))


@pytest.mark.parametrize("name", SYNTH)
def test_alt_against_golden(partls, name):
    g = load_golden(name)
    model, _, rep = partls.fit(partls.Alt, g["X"], g["y"], g["P"], η=float(g["eta"]), alpha0=g["alt_alpha0"], beta0=g["alt_beta0"])
    ref = float(g["alt_opt"])
    assert abs(rep.opt - ref) <= 1e-8 * max(1.0, ref)
    np.testing.assert_allclose(model.α, g["alt_alpha"], atol=1e-6)
    np.testing.assert_allclose(model.β, g["alt_beta"], atol=1e-6)
    assert abs(model.t - float(g["alt_t"])) < 1e-6


def test_toy_bnb_known_answer(partls):
    g = load_golden("toy")
    for X, y in [(g["X"], g["y"]), (g["X"].astype(np.float32), g["y"].astype(np.float32))]:
        result = partls.fit(partls.BnB, X, y, g["P"], η=0.0)
        assert abs(result[2].opt) < 1e-6
        y_pred = partls.predict(result[0], X)
        assert abs(np.sum(y_pred - g["y"]) ** 2) < 1e-6
        np.testing.assert_allclose(result[0].α, g["exact_alpha"], atol=1e-8)
        np.testing.assert_allclose(result[0].β, g["exact_beta"], atol=1e-8)
        assert result[2].nopen >= 1


@pytest.mark.parametrize("name", SYNTH + ["corr"])
def test_bnb_against_golden(partls, name):
    g = load_golden(name)
    model, _, rep = partls.fit(partls.BnB, g["X"], g["y"], g["P"], η=float(g.get("eta", 0.0)))
    ref = float(g["bnb_opt"])
    assert abs(rep.opt - ref) <= 1e-9 * max(1.0, ref)
    np.testing.assert_allclose(model.α, g["bnb_alpha"], atol=1e-7)
    np.testing.assert_allclose(model.β, g["bnb_beta"], atol=1e-7)
    assert abs(model.t - float(g["bnb_t"])) < 1e-7
    # identity: BnB optimum == Opt optimum
    assert abs(rep.opt - float(g["opt_opt"])) <= 1e-8 * max(1.0, float(g["opt_opt"]))


def test_bnb_equals_opt_mid_size(partls, oracle):
    X, y, P, _ = oracle.synth(20260120, 3000, 60, 10)
    mo, _, ro = partls.fit(partls.Opt, X, y, P)
    mb, _, rb = partls.fit(partls.BnB, X, y, P)
    assert abs(ro.opt - rb.opt) <= 1e-9 * max(1.0, ro.opt)
    np.testing.assert_allclose(mb.α, mo.α, atol=1e-7)
    np.testing.assert_allclose(mb.β, mo.β, atol=1e-7)
    assert rb.nopen < 2 ** 11                      # pruning works: far fewer nodes than patterns


def test_non_contiguous_partition_is_permuted_correctly(partls):
    """'corr' has a shuffled, unbalanced partition: exercises the group-contiguous variable permutation."""
    g = load_golden("corr")
    model, _, rep = partls.fit(partls.Opt, g["X"], g["y"], g["P"], faithful_intercept=True)
    assert rep.best_index == int(g["opt_best_index"])
    np.testing.assert_allclose(model.α, g["opt_alpha"], atol=1e-7)


def test_overlapping_partition_accepted_for_alt_bnb(partls, oracle):
    """any 0/1 P is valid input (PartitionedLS.jl:292); round 1 rejected overlapping groups for Alt / BnB — now they run on the
    device with per-variable node codes and agree with the oracle (more cases: test_gpu_configs.py)"""
    X = np.random.default_rng(0).standard_normal((50, 4)); y = X[:, 0] - 0.5 * X[:, 2] + 1
    P = np.array([[1, 0], [1, 1], [0, 1], [0, 1]])
    ref = oracle.fit_alt(X, y, P, np.ones(5), np.ones(3), eps=1e-6, T=100)
    m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=np.ones(5), beta0=np.ones(3))
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    rb = oracle.fit_bnb(X, y, P)
    m2, _, rep2 = partls.fit(partls.BnB, X, y, P)
    assert abs(rep2.opt - rb["opt"]) <= 1e-8 * max(1.0, rb["opt"])


def test_alt_and_opt_beyond_register_kernel(partls, oracle):
    """n = 321 > 320: enumeration on the global-memory kernel, single solves (Alt alpha-steps, winner re-solve) on the
    cooperative multi-workgroup kernel; both against the dense oracle."""
    X, y, P, _ = oracle.synth(20260150, 1500, 320, 4)
    rng = np.random.default_rng(5)
    a0 = rng.random(321); b0 = (rng.random(5) - 0.5) * 10
    ref = oracle.fit_alt(X, y, P, a0, b0)
    m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0)
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(m.α, ref["alpha"], atol=1e-6)
    np.testing.assert_allclose(m.β, ref["beta"], atol=1e-6)
    ro = oracle.fit_opt(X, y, P)
    mo, _, repo = partls.fit(partls.Opt, X, y, P)
    assert abs(repo.opt - ro["opt"]) <= 1e-9 * max(1.0, ro["opt"])
    np.testing.assert_allclose(mo.α, ro["alpha"], atol=1e-7)
    np.testing.assert_allclose(mo.β, ro["beta"], atol=1e-7)


def test_alt_determinism_like_the_reference_suite(partls):
    """runtests.jl:102-121 — ten fits of Alt with rng=123 on a 1000 x 10 regression with two groups of five give the same
    optimum (|Δ| <= 1e-6); make_regression is replaced by a fixed numpy draw (MLJBase is not available here)."""
    rng = np.random.default_rng(123)
    X = rng.standard_normal((1000, 10))
    y = X @ rng.standard_normal(10) + 0.1 * rng.standard_normal(1000)
    P = np.zeros((10, 2), dtype=np.int64); P[:5, 0] = 1; P[5:, 1] = 1
    last = None
    for _ in range(10):
        model, _, rep = partls.fit(partls.Alt, X, y, P, ϵ=1e-3, T=100, rng=123)
        y_pred = partls.predict(model, X)
        assert np.all(np.isfinite(y_pred))
        if last is None:
            last = rep.opt
        else:
            assert abs(rep.opt - last) <= 1e-6


def _branching_problem(seed=7, N=400, D=36, K=6):
    """a target the partitioned model cannot explain well: the relaxation at the root mixes signs in most groups, so BnB branches"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    y = X @ rng.standard_normal(D) + 0.2 * rng.standard_normal(N)            # unconstrained signs inside every group
    return X, y, P


@pytest.mark.parametrize("mode", ["warm", "cold", "tiny_pool", "batch7"])
def test_bnb_warm_started_bounds_equal_cold_ones(partls, oracle, monkeypatch, mode):
    """fit(BnB) with every node started from its parent's tableau snapshot (default), from the fresh tableau (PARTLS_BNB_COLD), with
    a snapshot pool too small for the frontier (children of unrecorded nodes start cold) and with odd batch sizes: same optimum and
    model as the oracle's depth-first search (BnB.jl:94-132) and as Opt."""
    if mode == "cold":
        monkeypatch.setenv("PARTLS_BNB_COLD", "1")
    if mode == "tiny_pool":
        monkeypatch.setenv("PARTLS_BNB_POOL_MB", "1")
    if mode == "batch7":
        monkeypatch.setenv("PARTLS_BNB_BATCH", "7")
    X, y, P = _branching_problem()
    ref = oracle.fit_bnb(X, y, P)
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    a, b, t, opt, nopen = ctx.bnb_prepared()
    mu, pat, free, nodes = ctx.bnb_search(0)
    ctx.close()
    assert nopen > 20 and nodes == nopen                                      # it really branched; the search is deterministic
    assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and abs(mu - opt) <= 1e-9 * max(1.0, opt)
    np.testing.assert_allclose(partls.predict(partls.PartLSFitResult(a, b, t, P), X),
                               oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]), atol=1e-7)
    m, _, rep = partls.fit(partls.Opt, X, y, P)
    assert abs(rep.opt - opt) <= 1e-9 * max(1.0, opt)


def test_bnb_search_node_cap(partls):
    X, y, P = _branching_problem(seed=8, D=48, K=8)
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    mu_full, _, _, n_full = ctx.bnb_search(0)
    mu_cap, _, _, n_cap = ctx.bnb_search(5)
    ctx.close()
    assert 5 <= n_cap <= 5 + 1024 and n_cap <= n_full and (mu_cap >= mu_full)


def _many_groups_problem(K=45, seed=31):
    """K groups (> 39: beyond the enumeration of Opt, which needs 2^K solves): 5 groups of four features with mixed true signs — the
    only ones BnB can branch on — and K - 5 singletons"""
    rng = np.random.default_rng(seed)
    D = 20 + (K - 5)
    P = np.zeros((D, K), dtype=np.int64)
    for g in range(5):
        P[4 * g:4 * g + 4, g] = 1
    for j in range(K - 5):
        P[20 + j, 5 + j] = 1
    N = 600
    X = rng.standard_normal((N, D))
    y = X @ rng.standard_normal(D) + 0.4 + 0.1 * rng.standard_normal(N)
    return X, y, P


def test_alt_and_bnb_take_more_groups_than_opt_can_enumerate(partls, oracle):
    """The reference has no limit on the number of groups; Opt's 2^K enumeration stops at K = 39 here, Alt and BnB (64-bit group masks)
    at K = 61 — with a clear status for fit(Opt) beyond its range."""
    X, y, P = _many_groups_problem()
    K = P.shape[1]
    rng = np.random.default_rng(3)
    a0 = rng.random(X.shape[1] + 1); b0 = (rng.random(K + 1) - 0.5) * 10
    ref = oracle.fit_alt(X, y, P, a0, b0, T=30)
    m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0, T=30)
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(partls.predict(m, X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]), atol=1e-6)
    refb = oracle.fit_bnb(X, y, P)
    mb, _, repb = partls.fit(partls.BnB, X, y, P)
    assert abs(repb.opt - refb["opt"]) <= 1e-9 * max(1.0, refb["opt"]) and repb.nopen >= 3
    np.testing.assert_allclose(partls.predict(mb, X), oracle.predict(X, P, refb["alpha"], refb["beta"], refb["t"]), atol=1e-7)
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.Opt, X, y, P)
    assert ei.value.status == partls.lowlevel.ERR_UNSUPPORTED and "K <= 39" in str(ei.value)
    Pbig = np.zeros((X.shape[1], 62), dtype=np.int64); Pbig[np.arange(X.shape[1]), np.arange(X.shape[1]) % 62] = 1
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.BnB, X, y, Pbig)
    assert ei.value.status == partls.lowlevel.ERR_UNSUPPORTED


def test_bnb_search_warm_through_the_snapshot_abi(partls, oracle):
    """dist.bnb_search_warm on the real context (partls_bnb_snap_begin / _bound_snap / _snap_release): the host-driven search with
    snapshot slots equals the in-library search (same optimum, same node count for the same batch size) and the oracle."""
    X, y, P = _branching_problem(seed=9, D=40, K=7)
    ref = oracle.fit_bnb(X, y, P)
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    mu_lib, pat_lib, free_lib, n_lib = ctx.bnb_search(0)
    mu, pat, free, n = partls.dist.bnb_search_warm(ctx, P.shape[1] + 1, batch=1024)
    cold = partls.dist.bnb_search(ctx.bnb_bound, P.shape[1] + 1)
    a, b, t, opt = ctx.bnb_leaf(pat, free)
    ctx.close()
    assert abs(mu - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and abs(mu - mu_lib) <= 1e-12 * mu_lib and abs(mu - cold[0]) <= 1e-10 * mu
    assert (pat, free) == (pat_lib, free_lib) and n == n_lib and n > 20
    assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


@pytest.mark.parametrize("D", [290, 312])
def test_alt_bnb_and_opt_between_the_two_sweep_kernels(partls, oracle, D):
    """n = 291 / 313: since round 4 (second half) these sizes run on the deferred-update kernel and, for single solves, on the cooperative
    kernel — no longer on the register kernel's spilling T = 19 / 20 instantiations.  Alt, Opt and a BnB search that branches (snapshots
    of the deferred-update kernel) against the dense oracle."""
    X, y, P, _ = oracle.synth(20260160 + D, 1200, D, 4)
    rng = np.random.default_rng(D)
    a0 = rng.random(D + 1); b0 = (rng.random(5) - 0.5) * 10
    ref = oracle.fit_alt(X, y, P, a0, b0)
    m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0)
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(m.α, ref["alpha"], atol=1e-6)
    ro = oracle.fit_opt(X, y, P)
    mo, _, repo = partls.fit(partls.Opt, X, y, P)
    assert abs(repo.opt - ro["opt"]) <= 1e-9 * max(1.0, ro["opt"])
    np.testing.assert_allclose(mo.α, ro["alpha"], atol=1e-7)
    yn = 1.0 + 0.3 * np.random.default_rng(D + 1).standard_normal(X.shape[0])       # a target nobody explains: the search has to branch
    rb = oracle.fit_bnb(X, yn, P)
    mb, _, repb = partls.fit(partls.BnB, X, yn, P)
    assert abs(repb.opt - rb["opt"]) <= 1e-8 * max(1.0, rb["opt"])
    mo2, _, repo2 = partls.fit(partls.Opt, X, yn, P)
    assert abs(repb.opt - repo2.opt) <= 1e-8 * max(1.0, repo2.opt)
