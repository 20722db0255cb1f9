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
