"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the golden fixtures,
the reference's known answers and the CPU oracle.  Tolerance: |opt_gpu - opt_ref| <= 1e-9 * max(1, opt_ref) per pattern
and for the optimum; model (alpha, beta, t) within 1e-7."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

SYNTH = ["synth_a", "synth_b", "synth_eta", "synth_c", "corr"]
TOL_OBJ = 1e-9
TOL_MODEL = 1e-7


def test_hip_library_is_the_thing_that_runs(partls):
    import os
    assert os.path.exists(partls.library_path())
    assert partls.lowlevel.lib().partls_device_count() >= 1


@pytest.mark.parametrize("generic", [True, False])
def test_toy_standard_interface(partls, generic):
    """test/runtests.jl:8-39 — result = fit(Opt, X, y, P, η=0.0); opt ≈ 0 atol 1e-6; sum(ŷ - y)^2 ≈ 0 atol 1e-6."""
    g = load_golden("toy")
    result = partls.fit(partls.Opt, g["X"], g["y"], g["P"], η=0.0, generic_kernel=generic)
    opt = result[2].opt
    y_pred = partls.predict(result[0], g["X"])
    assert abs(opt) < 1e-6
    assert abs(np.sum(y_pred - g["y"]) ** 2) < 1e-6
    np.testing.assert_allclose(result[0].α, g["exact_alpha"], atol=1e-9)
    np.testing.assert_allclose(result[0].β, g["exact_beta"], atol=1e-9)
    assert abs(result[0].t - float(g["exact_t"])) < 1e-9


@pytest.mark.parametrize("generic", [True, False])
def test_toy_float32(partls, generic):
    """test/runtests.jl:123-146 — Float32 inputs."""
    g = load_golden("toy")
    result = partls.fit(partls.Opt, g["X"].astype(np.float32), g["y"].astype(np.float32), g["P"], η=0.0, generic_kernel=generic)
    assert abs(result[2].opt) < 1e-6
    y_pred = partls.predict(result[0], g["X"].astype(np.float32))
    assert abs(np.sum(y_pred - g["y"]) ** 2) < 1e-6


@pytest.mark.parametrize("generic", [True, False])
def test_toy_all_patterns_faithful(partls, generic):
    """all 2^(K+1) optvals (Opt.jl:90) and winner b = 5, first-index argmin (Opt.jl:96)."""
    g = load_golden("toy")
    model, _, rep = partls.fit(partls.Opt, g["X"], g["y"], g["P"], returnAllSolutions=True, generic_kernel=generic)
    sols = rep.solutions
    assert len(sols) == 8
    objs = np.array([sols._all[b] for b in range(8)])
    np.testing.assert_allclose(objs, g["opt_all_opt"], atol=2e-7)     # Gram-form optval: abs error ~ sqrt(eps*yy) at obj = 0
    np.testing.assert_allclose(np.delete(objs, 5), np.delete(g["opt_all_opt"], 5), atol=1e-11)
    for b in range(8):
        o, m = sols[b]
        np.testing.assert_allclose(m.α, g["opt_all_alpha"][b], atol=1e-9)
        np.testing.assert_allclose(m.β, g["opt_all_beta"][b], atol=1e-9)
        assert abs(m.t - g["opt_all_t"][b]) < 1e-9


@pytest.mark.parametrize("generic", [True, False])
@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("name", SYNTH)
def test_opt_against_golden(partls, name, faithful, generic):
    g = load_golden(name)
    eta = float(g.get("eta", 0.0))
    model, _, rep = partls.fit(partls.Opt, g["X"], g["y"], g["P"], η=eta, faithful_intercept=faithful, generic_kernel=generic)
    ref = float(g["opt_opt"])
    assert abs(rep.opt - ref) <= TOL_OBJ * max(1.0, ref)
    assert rep.best_index == int(g["opt_best_index"])
    np.testing.assert_allclose(model.α, g["opt_alpha"], atol=TOL_MODEL)
    np.testing.assert_allclose(model.β, g["opt_beta"], atol=TOL_MODEL)
    assert abs(model.t - float(g["opt_t"])) < TOL_MODEL


@pytest.mark.parametrize("generic", [True, False])
@pytest.mark.parametrize("name", SYNTH)
def test_every_pattern_objective_against_golden(partls, name, generic):
    """per-pattern parity: all 2^(K+1) optvals of the sweep vs the scipy fixture"""
    g = load_golden(name)
    ctx = partls.default_context()
    flags = partls.lowlevel.OPT_FAITHFUL_INTERCEPT | (partls.lowlevel.OPT_GENERIC_KERNEL if generic else 0)
    ctx.opt_prepare(g["X"], g["y"], g["P"], float(g.get("eta", 0.0)), flags)
    bo, bp, allopt, unconv = ctx.opt_sweep(0, -1, want_all=True)
    assert unconv == 0
    ref = g["opt_all_opt"]
    np.testing.assert_allclose(allopt, ref, rtol=0, atol=TOL_OBJ * max(1.0, ref.max()))
    assert bp == int(g["opt_best_index"])


def test_gram_build_matches_numpy(partls, oracle):
    X, y, P, _ = oracle.synth(20260110, 3000, 70, 7)
    ctx = partls.default_context()
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    G = ctx.gram()
    Z = np.hstack([X, np.ones((3000, 1)), y[:, None]])
    ref = Z.T @ Z
    np.testing.assert_allclose(G, ref, rtol=1e-12, atol=1e-9)
    assert np.array_equal(G, G.T)


def test_device_generator_bit_identical_to_host(partls, oracle):
    import torch
    N, D, K = 4096 + 37, 19, 4
    seed = 20260111
    P, ws = partls.synth_truth(seed, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device="cuda")
    dy = torch.empty(N, dtype=torch.float64, device="cuda")
    ctx = partls.default_context()
    ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    Xh, yh, Ph, wh = oracle.synth(seed, N, D, K)
    assert np.array_equal(dX.cpu().numpy().reshape(D, N).T, Xh)
    assert np.array_equal(dy.cpu().numpy(), yh)
    assert np.array_equal(P, Ph)


@pytest.mark.parametrize("generic", [True, False])
def test_mid_size_sampled_patterns_vs_oracle(partls, oracle, generic):
    """N=4000, D=48, K=8: all 2^9 optvals vs the oracle on the QR-compressed data (identical objective for every w)."""
    X, y, P, _ = oracle.synth(20260112, 4000, 48, 8)
    ctx = partls.default_context()
    flags = partls.lowlevel.OPT_FAITHFUL_INTERCEPT | (partls.lowlevel.OPT_GENERIC_KERNEL if generic else 0)
    ctx.opt_prepare(X, y, P, 0.0, flags)
    bo, bp, allopt, unconv = ctx.opt_sweep(0, -1, want_all=True)
    assert unconv == 0
    Xo, Po = oracle.homogeneous(X, P)
    R, z = oracle.compress(Xo, y)
    ref = oracle.opt_patterns(R, z, Po, np.arange(512))
    np.testing.assert_allclose(allopt, ref, rtol=1e-10)
    assert bp == int(np.argmin(ref))
    a, b, t, opt, bi = ctx.opt_finish(bp)
    ro = oracle.fit_opt(X, y, P)
    assert abs(opt - ro["opt"]) <= TOL_OBJ * max(1.0, ro["opt"])
    np.testing.assert_allclose(a, ro["alpha"], atol=TOL_MODEL)
    np.testing.assert_allclose(b, ro["beta"], atol=TOL_MODEL)


def test_shards_compose(partls, oracle):
    """multi-GPU sharding logic on one device: per-shard minima reduce to the full-sweep result (lexicographic min)."""
    X, y, P, _ = oracle.synth(20260113, 1500, 30, 6)
    ctx = partls.default_context()
    ctx.opt_prepare(X, y, P, 0.0, 0)
    npat = ctx.num_patterns()
    full = ctx.opt_sweep(0, -1)
    parts = [ctx.opt_sweep(r * npat // 4, (r + 1) * npat // 4) for r in range(4)]
    best = min((p[0], p[1]) for p in parts)
    assert best == (full[0], full[1])


def test_errors(partls):
    X = np.array([[1., 2, 3], [3, 3, 4], [8, 1, 3], [5, 3, 1]]); y = np.array([1., 1, 2, 3])
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.Opt, X, y, np.array([[1, 0], [2, 0], [0, 1]]))
    assert ei.value.status == partls.lowlevel.ERR_BAD_PARTITION
    Xn = X.copy(); Xn[1, 1] = np.nan
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.Opt, Xn, y, np.array([[1, 0], [1, 0], [0, 1]]))
    assert ei.value.status == partls.lowlevel.ERR_NONFINITE
    for bad_y in (np.inf, -np.inf, np.nan):                      # the screen is the Gram diagonal (sum of squares per column of [X y])
        yn = y.copy(); yn[2] = bad_y
        with pytest.raises(partls.PartlsError) as ei:
            partls.fit(partls.Opt, X, yn, np.array([[1, 0], [1, 0], [0, 1]]))
        assert ei.value.status == partls.lowlevel.ERR_NONFINITE
    Xi = X.copy(); Xi[3, 0] = -np.inf
    with pytest.raises(partls.PartlsError) as ei:
        partls.fit(partls.Alt, Xi, y, np.array([[1, 0], [1, 0], [0, 1]]))
    assert ei.value.status == partls.lowlevel.ERR_NONFINITE
    # limits of this build are refused up front, before any memory is touched: ldX < 2^30 (32-bit element offsets in the Gram
    # kernel's loads), M <= 1022, K <= 39
    import torch
    d = torch.zeros(16, dtype=torch.float64, device="cuda")
    ctx = partls.Context()
    with pytest.raises(partls.PartlsError) as ei:
        ctx.opt_prepare_device(d.data_ptr(), d.data_ptr(), 4, 3, 1 << 30, np.array([[1, 0], [1, 0], [0, 1]]))
    assert ei.value.status == partls.lowlevel.ERR_UNSUPPORTED
    with pytest.raises(partls.PartlsError) as ei:
        ctx.opt_prepare_device(d.data_ptr(), d.data_ptr(), 4, 3, 3, np.array([[1, 0], [1, 0], [0, 1]]))          # ldX < N
    assert ei.value.status == partls.lowlevel.ERR_BAD_ARG
    ctx.close()


# ---- visiting order of the sweep: which group sits on which Gray bit (calibrated on long enumerations) ---------------------------
def _ctx_with_order(partls, monkeypatch, mode):
    """a private context whose PARTLS_BIT_ORDER knob (read once at partls_create) is `mode`"""
    monkeypatch.setenv("PARTLS_BIT_ORDER", mode)
    return partls.Context(0)


def _order_problem(seed=11, N=400, M=45, K=9, empty_group=None):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, M))
    P = np.zeros((M, K), dtype=np.int64)
    groups = [k for k in range(K) if k != empty_group]
    P[np.arange(M), np.array(groups)[rng.integers(0, len(groups), size=M)]] = 1
    sizes = P.sum(axis=0)
    w = rng.standard_normal(M) * (rng.random(M) < 0.6)
    y = X @ w + 0.5 + 0.3 * rng.standard_normal(N)
    return np.asfortranarray(X), y, np.asfortranarray(P), sizes


def test_calibrated_order_visits_the_same_subproblems(partls, oracle, monkeypatch):
    """The sweep may put any group on any Gray bit (it measures the flip cost of every group and gives the cheap ones the fast bits);
    the boundary keeps the reference's pattern index (Opt.jl:4-12): per-pattern objectives, winner and model are those of the plain
    order and of the oracle, and shards of the Gray-index space still cover every pattern exactly once."""
    X, y, P, _ = _order_problem()
    flags = partls.lowlevel.OPT_FAITHFUL_INTERCEPT
    res = {}
    for mode in ("identity", "calibrate"):
        ctx = _ctx_with_order(partls, monkeypatch, mode)
        ctx.opt_prepare(X, y, P, 0.0, flags)
        gbit, cost = ctx.bit_order()
        npat = ctx.num_patterns()
        bo, bp, allo, unconv = ctx.opt_sweep(0, -1, want_all=True)
        assert unconv == 0
        parts = [ctx.opt_sweep(*partls.dist.shard_range(npat, r, 3), want_all=True) for r in range(3)]
        assert min((p[0], p[1]) for p in parts)[1] == bp
        # every pattern is visited by exactly one shard (entries outside a shard are NaN), with the full sweep's value up to the
        # round-off of a different chain path
        seen = np.stack([~np.isnan(p[2]) for p in parts])
        assert np.all(seen.sum(axis=0) == 1)
        assert [int(v.sum()) for v in seen] == [b - a for a, b in (partls.dist.shard_range(npat, r, 3) for r in range(3))]
        np.testing.assert_allclose(np.nansum(np.stack([p[2] for p in parts]), axis=0), allo, rtol=1e-9, atol=1e-10)
        res[mode] = dict(gbit=gbit, cost=cost, bo=bo, bp=bp, allo=allo.copy(), model=ctx.opt_finish(bp))
        ctx.close()
    K1 = P.shape[1] + 1
    assert list(res["identity"]["gbit"]) == list(range(K1)) and np.all(res["identity"]["cost"] == -1.0)
    g = res["calibrate"]["gbit"]
    assert sorted(g) == list(range(K1)) and list(g) != list(range(K1))         # a real permutation on this problem
    c = res["calibrate"]["cost"]
    assert np.all(c >= 0) and np.all(np.diff(c[np.argsort(g)]) >= 0)          # cheaper groups sit on faster bits
    np.testing.assert_allclose(res["calibrate"]["allo"], res["identity"]["allo"], rtol=1e-9, atol=1e-10)
    assert res["calibrate"]["bp"] == res["identity"]["bp"]
    for u, v in zip(res["calibrate"]["model"], res["identity"]["model"]):
        np.testing.assert_allclose(u, v, rtol=1e-9, atol=1e-10)
    ref = oracle.fit_opt(X, y, P, 0.0, return_all=True)
    np.testing.assert_allclose(res["calibrate"]["allo"], ref["all_opt"], rtol=1e-8, atol=1e-9)
    assert res["calibrate"]["model"][4] == ref["best_index"]


def test_calibrated_order_empty_group_and_first_index_tie(partls, oracle, monkeypatch):
    """A group without features costs nothing to flip, so the calibration gives it the fastest bit; its two patterns tie exactly and
    best_index must still be the reference's first index (Opt.jl:96), whatever the internal order."""
    X, y, P, sizes = _order_problem(seed=12, K=8, empty_group=5)
    assert sizes[5] == 0
    ctx = _ctx_with_order(partls, monkeypatch, "calibrate")
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    gbit, cost = ctx.bit_order()
    assert cost[5] == 0.0 and gbit[5] == 0
    bo, bp, allo, _ = ctx.opt_sweep(0, -1, want_all=True)
    a, b, t, opt, bi = ctx.opt_finish(bp)
    ctx.close()
    ref = oracle.fit_opt(X, y, P, 0.0, return_all=True)
    assert bi == ref["best_index"] and not (bi >> 5) & 1
    np.testing.assert_allclose(allo, ref["all_opt"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(allo[np.arange(len(allo)) | (1 << 5)], allo[np.arange(len(allo)) & ~(1 << 5)], rtol=1e-12)


def test_calibrated_order_on_the_global_memory_kernel(partls, oracle, monkeypatch):
    """n = 331 > 320 tableau variables: the sweep and the calibration's chains of nodes run on sweep_generic.hip"""
    X, y, P, _ = _order_problem(seed=13, N=700, M=330, K=5)
    ctx = _ctx_with_order(partls, monkeypatch, "calibrate")
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    gbit, cost = ctx.bit_order()
    assert sorted(gbit) == list(range(6)) and np.all(cost >= 0) and np.all(np.diff(cost[np.argsort(gbit)]) >= 0)
    bo, bp, allo, unconv = ctx.opt_sweep(0, -1, want_all=True)
    a, b, t, opt, bi = ctx.opt_finish(bp)
    ctx.close()
    ref = oracle.fit_opt(X, y, P, 0.0, return_all=True)
    assert unconv == 0 and bi == ref["best_index"]
    np.testing.assert_allclose(allo, ref["all_opt"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(opt, ref["opt"], rtol=1e-9)


def test_tie_between_null_column_groups_keeps_the_first_reference_index(partls, oracle, monkeypatch):
    """Two groups that own only all-zero feature columns (non-empty groups, Opt.jl:28-29 gives them +-0 columns): their four sign
    combinations are the same subproblem, the sweep sees bitwise ties INSIDE a workgroup's chain (the calibration puts both groups
    on the two fastest Gray bits, in an order of its own), and both the raw sweep winner — what dist.allreduce_argmin receives —
    and fit's best_index must be the reference's first index (argmin, Opt.jl:96) in either visiting order."""
    X, y, P, _ = _order_problem(seed=21, K=7)
    K = P.shape[1]
    X = np.asfortranarray(np.hstack([X, np.zeros((X.shape[0], 3))]))
    P2 = np.zeros((X.shape[1], K + 2), dtype=np.int64)
    # the two null groups get the reference bits 1 and K+1; the real groups keep their relative order around them
    cols = [0, 2, 3, 4, 5, 6, 7]
    assert K == len(cols)
    P2[:P.shape[0], cols] = P
    P2[P.shape[0], 1] = 1; P2[P.shape[0] + 1, 1] = 1; P2[P.shape[0] + 2, K + 1] = 1
    ref = oracle.fit_opt(X, y, P2, 0.0, return_all=True)
    nullmask = (1 << 1) | (1 << (K + 1))
    assert ref["best_index"] & nullmask == 0
    for mode in ("identity", "calibrate"):
        ctx = _ctx_with_order(partls, monkeypatch, mode)
        ctx.opt_prepare(X, y, P2, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
        gbit, cost = ctx.bit_order()
        if mode == "calibrate":
            assert cost[1] == 0.0 and cost[K + 1] == 0.0 and sorted([gbit[1], gbit[K + 1]]) == [0, 1]
        bo, bp, allo, unconv = ctx.opt_sweep(0, -1, want_all=True)
        assert unconv == 0
        assert bp == int(np.argmin(allo)), (mode, bp, int(np.argmin(allo)))      # first index among the bitwise-equal minima
        assert bp & nullmask == 0
        # shards whose boundaries cut through the tie set: members of the set that sit in different shards are reached along different
        # chain paths, so their tracked objectives agree to round-off only and any of them may win the reduce — finish() then clears
        # the bits of the null groups, which is what makes best_index the reference's
        parts = [ctx.opt_sweep(*partls.dist.shard_range(ctx.num_patterns(), r, 5)) for r in range(5)]
        bps = min((p[0], p[1]) for p in parts)[1]
        assert bps & ~nullmask == bp
        assert ctx.opt_finish(bps)[4] == ref["best_index"]
        a, b, t, opt, bi = ctx.opt_finish(bp)
        ctx.close()
        assert bi == ref["best_index"], (mode, bi, ref["best_index"])
        np.testing.assert_allclose(allo, ref["all_opt"], rtol=1e-8, atol=1e-9)
    # free-intercept mode through fit(): the null groups' bits stay clear as well
    m, _, rep = partls.fit(partls.Opt, X, y, P2)
    assert rep.best_index == ref["best_index"]


def test_solutions_survive_a_later_fit(partls):
    """returnAllSolutions (Opt.jl:99-101): the returned (opt_b, model_b) list stays valid after other fits have used the shared
    context, as in the reference — the models are rebuilt on a private context once the shared one has moved on."""
    g = load_golden("toy")
    model, _, rep = partls.fit(partls.Opt, g["X"], g["y"], g["P"], returnAllSolutions=True)
    sols = rep.solutions
    first = sols[3]
    rng = np.random.default_rng(0)
    X2 = rng.standard_normal((50, 5)); y2 = rng.standard_normal(50); P2 = np.eye(5, dtype=np.int64)[:, :2]; P2[2:, 1] = 1
    partls.fit(partls.Opt, X2, y2, P2)                                   # takes the shared context over
    for b in range(8):
        o, m = sols[b]
        np.testing.assert_allclose(m.α, g["opt_all_alpha"][b], atol=1e-9)
        np.testing.assert_allclose(m.β, g["opt_all_beta"][b], atol=1e-9)
    np.testing.assert_array_equal(sols[3][1].α, first[1].α)


def test_alt_feature_in_no_group_is_ignored_not_nan(partls, oracle):
    """A feature whose row of P is zero never enters the model (predict multiplies it by 0).  Alt.jl:98 divides 0 / 0 for it; here
    its alpha stays 0 and the fit equals the fit without that column."""
    rng = np.random.default_rng(4)
    N, M, K = 300, 7, 3
    X = rng.standard_normal((N, M)); y = X[:, :6] @ rng.standard_normal(6) + 0.3 + 0.1 * rng.standard_normal(N)
    P = np.zeros((M, K), dtype=np.int64); P[[0, 1], 0] = 1; P[[2, 3], 1] = 1; P[[4, 5], 2] = 1       # feature 6: no group
    a0 = rng.random(M + 1); b0 = (rng.random(K + 1) - 0.5) * 10
    m, _, rep = partls.fit(partls.Alt, X, y, P, alpha0=a0, beta0=b0, T=50)
    assert np.all(np.isfinite(m.α)) and np.all(np.isfinite(m.β)) and np.isfinite(rep.opt) and m.α[6] == 0.0
    ref = oracle.fit_alt(X[:, :6], y, P[:6], np.delete(a0, 6), b0, T=50)
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(m.α[:6], ref["alpha"], atol=1e-7)


def test_winner_solution_left_by_the_sweep_equals_a_fresh_solve(partls, oracle, monkeypatch):
    """Small tableaus (256-thread kernel): partls_opt_finish starts from the solution the sweep left for its winner instead of solving
    that pattern again; PARTLS_NO_EXPORT=1 forces the re-solve.  Same winner, objective and model, in both intercept modes; a sharded
    sweep (the winner may come from another shard: no export to use) agrees too."""
    X, y, P, _ = _order_problem(seed=31, N=900, M=120, K=10)
    ref = oracle.fit_opt(X, y, P)
    res = {}
    for mode in ("export", "resolve"):
        if mode == "resolve":
            monkeypatch.setenv("PARTLS_NO_EXPORT", "1")
        else:
            monkeypatch.delenv("PARTLS_NO_EXPORT", raising=False)
        ctx = partls.Context(0)
        for flags in (0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT):
            ctx.opt_prepare(X, y, P, 0.0, flags)
            bo, bp, _, unconv = ctx.opt_sweep(0, -1)
            res[mode, flags] = ctx.opt_finish(bp)
            assert unconv == 0
        npat = ctx.num_patterns()
        parts = [ctx.opt_sweep(*partls.dist.shard_range(npat, r, 2)) for r in range(2)]
        res[mode, "sharded"] = ctx.opt_finish(min((p[0], p[1]) for p in parts)[1])
        ctx.close()
    for key in [k for k in res if k[0] == "export"]:
        a, b, t, opt, bi = res[key]
        a2, b2, t2, opt2, bi2 = res["resolve", key[1]]
        assert bi == bi2 and abs(opt - opt2) <= 1e-12 * max(1.0, opt2) and abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        np.testing.assert_allclose(a, a2, atol=1e-9)
        np.testing.assert_allclose(b, b2, atol=1e-9)
        assert abs(t - t2) <= 1e-9
