"""GPU parity tests of the experimental two-level sweep kernel (sweep_two.hip, opt-in with PARTLS_KERNEL=two): every
pattern's objective must agree with the production single-level kernel and with the golden fixtures; the discovery and the
classical-fallback paths are forced through the PARTLS_LOW_ECAP test hook.  Tolerance as in test_gpu_opt.py: 1e-9 relative
per pattern (both kernels solve the same KKT systems; they differ only in pivot order)."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL_OBJ = 1e-9


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _partition(D, K):
    P = np.zeros((D, K), dtype=np.int64)
    sizes = [D // K + (1 if k < D % K else 0) for k in range(K)]
    c = 0
    for k, s in enumerate(sizes):
        P[c:c + s, k] = 1
        c += s
    return P


def _problem(N, D, K, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    P = _partition(D, K)
    grp = P.argmax(1)
    a = rng.random(D)
    for k in range(K):
        a[grp == k] /= a[grp == k].sum()
    beta = (rng.random(K) - 0.5) * 10
    y = X @ (a * beta[grp]) + 1.0 + noise * rng.standard_normal(N)
    return X, y, P


def _all_patterns(partls, X, y, P, **env):
    with _Env(**env):
        ctx = partls.Context()
        ctx.opt_prepare(X, y, P, 0.0, 1)                       # faithful intercept: per-pattern objectives are defined
        obj, pat, allo, unc = ctx.opt_sweep(0, ctx.num_patterns(), want_all=True)
        piv = ctx.pivots()
    return obj, pat, allo, unc, piv


@pytest.mark.parametrize("shape", [(3000, 48, 6, 11), (4000, 100, 9, 12), (5000, 130, 10, 13), (6000, 250, 7, 14)])
def test_two_level_matches_single_level_every_pattern(partls, shape):
    N, D, K, seed = shape
    X, y, P = _problem(N, D, K, seed)
    o1, p1, a1, u1, piv1 = _all_patterns(partls, X, y, P, PARTLS_KERNEL="blk")
    o2, p2, a2, u2, piv2 = _all_patterns(partls, X, y, P, PARTLS_KERNEL="two")
    assert u1 == 0 and u2 == 0
    assert p1 == p2
    np.testing.assert_allclose(a2, a1, rtol=TOL_OBJ, atol=1e-12)
    if D // K <= 15:                                               # two leading groups fit tile columns 0-1: two-level is active
        assert piv2 < piv1                                         # the point of the exercise: fewer pivots on the big tableau
    else:
        assert piv2 == piv1                                        # groups too large: the launcher stays on the single-level kernel


@pytest.mark.parametrize("ecap", [0, 1, 3])
def test_two_level_fallback_and_discovery_paths(partls, ecap):
    """ecap = 0: every block that needs a discovery is finished classically; 1 / 3: mixtures of discovery and overflow."""
    X, y, P = _problem(5000, 120, 8, 21)
    o1, p1, a1, u1, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="blk")
    o2, p2, a2, u2, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="two", PARTLS_LOW_ECAP=ecap)
    assert u2 == 0 and p1 == p2
    np.testing.assert_allclose(a2, a1, rtol=TOL_OBJ, atol=1e-12)


def test_two_level_one_low_group_and_chain_alignment(partls):
    """PARTLS_LOW_GROUPS=1 (two patterns per block) and a chain length that is not a multiple of the block: partial blocks run
    classically."""
    X, y, P = _problem(4000, 96, 8, 31)
    o1, p1, a1, u1, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="blk")
    o2, p2, a2, u2, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="two", PARTLS_LOW_GROUPS=1)
    np.testing.assert_allclose(a2, a1, rtol=TOL_OBJ, atol=1e-12)
    with _Env(PARTLS_KERNEL="two"):
        ctx = partls.Context()
        ctx.opt_prepare(X, y, P, 0.0, 1)
        n = ctx.num_patterns()
        oa, pa, _, _ = ctx.opt_sweep(3, n - 5)                      # unaligned shard of the Gray range
    lo = np.array(a1)
    sub = lo[[(g ^ (g >> 1)) for g in range(3, n - 5)]]
    assert abs(oa - sub.min()) <= TOL_OBJ * max(1.0, sub.min())


@pytest.mark.parametrize("name", ["synth_a", "synth_c", "corr"])
def test_two_level_against_golden(partls, name):
    g = load_golden(name)
    with _Env(PARTLS_KERNEL="two"):
        model, _, rep = partls.fit(partls.Opt, g["X"], g["y"], g["P"], η=float(g["eta"]) if "eta" in g else 0.0)
    assert abs(rep.opt - float(g["opt_opt"])) <= TOL_OBJ * max(1.0, float(g["opt_opt"]))


def test_two_level_rank_deficient_and_duplicates(partls):
    """dependent columns inside the low groups and among the frozen variables (Lawson–Hanson rejection in both tableaus)."""
    X, y, P = _problem(3000, 80, 8, 41)
    X[:, 3] = X[:, 1]                                              # duplicate inside low group 0
    X[:, 15] = 2.0 * X[:, 12]                                      # dependent pair inside low group 1
    X[:, 60] = X[:, 50] - X[:, 55]                                 # dependent triple among the frozen variables
    X[:, 70] = 0.0                                                 # null column
    o1, p1, a1, u1, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="blk")
    o2, p2, a2, u2, _ = _all_patterns(partls, X, y, P, PARTLS_KERNEL="two")
    assert u2 == 0
    np.testing.assert_allclose(a2, a1, rtol=1e-8, atol=1e-10)


def test_two_level_full_c3_sweep_same_winner(partls):
    """BASELINE config 3 (N=100k, D=256, K=20, 2^20 patterns, device-generated data): both kernels find the same pattern and
    objective; the two-level kernel needs far fewer pivots on the register tableau."""
    import torch
    N, D, K, seed = 100000, 256, 20, 20260003
    P, wstar = partls.synth_truth(seed, D, K)
    dev = torch.device("cuda:0")
    dX = torch.empty(N * D, dtype=torch.float64, device=dev)
    dy = torch.empty(N, dtype=torch.float64, device=dev)
    res = {}
    for kern in ("blk", "two"):
        with _Env(PARTLS_KERNEL=kern):
            ctx = partls.Context()
            ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
            ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
            obj, pat, _, unc = ctx.opt_sweep(0, ctx.num_patterns())
            res[kern] = (obj, pat, unc, ctx.pivots())
    assert res["blk"][2] == 0 and res["two"][2] == 0
    assert res["blk"][1] == res["two"][1]
    assert abs(res["blk"][0] - res["two"][0]) <= 1e-10 * res["blk"][0]
    assert res["two"][3] < 0.6 * res["blk"][3]
