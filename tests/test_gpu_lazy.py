"""GPU tests of the deferred-update sweep kernel (csrc/sweep_lazy.hip): fit(Opt) beyond n = 320 tableau variables, and the node batches
of BnB / the bit-order calibration there (Opt.jl:87-90 per pattern; BnB.jl:69-92 per node).

The kernel keeps the rank-1 terms of its pivots pending in LDS and brings the tableau in global memory up to date only every ~40 pivots,
so the tests aim at what that adds: many flushes per chain (long chains), columns that are replaced and re-entered inside one pending
window (few groups, many flips), refused pivots (exactly dependent columns: the veto redo of the two-phase panel), every LDS plan
(512 threads up to n = 511, 1024 beyond), and node mode.  References: the oracle (dense Lawson-Hanson on QR-compressed data) for sampled
patterns, and the eager kernel (sweep_generic.hip, PARTLS_EAGER_GENERIC=1: every block applied to the whole tableau at once) for ALL
patterns — both kernels take the same decisions, so their objectives agree to round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(seed, N, D, K, dup=0, noise=0.3, trip=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    for i in range(dup):                                            # exactly dependent columns: copies and a sum of two
        X[:, D - 1 - i] = X[:, i] if i % 2 == 0 else X[:, i] + X[:, i + 1]
    for t in range(trip):                                           # x_i = 0.5 x_j - 2 x_l with a small x_j: x_i and x_l nearly collinear on the
        l, j, i = 20 + 3 * t, 21 + 3 * t, 22 + 3 * t                # unit scale, x_j exactly dependent on the pair with coefficients of a few hundred —
        X[:, j] *= 0.004                                            # its Gram-form pivot comes out above 1e-11 and only the leave-one-out rule
        X[:, i] = 0.5 * X[:, j] - 2.0 * X[:, l]                     # refuses it (tests/test_gpu_fuzz.py: the round-1 regressions)
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), rng.integers(0, K, size=D)] = 1
    w = rng.standard_normal(D) * (rng.random(D) < 0.6)
    y = X @ w + 0.4 + noise * rng.standard_normal(N)
    return np.asfortranarray(X), y, np.asfortranarray(P)


def _sweep(partls, monkeypatch, X, y, P, eager, chain_len=None, flags=None):
    if eager:
        monkeypatch.setenv("PARTLS_EAGER_GENERIC", "1")
    else:
        monkeypatch.delenv("PARTLS_EAGER_GENERIC", raising=False)
    if chain_len:
        monkeypatch.setenv("PARTLS_CHAIN_LEN", str(chain_len))
    else:
        monkeypatch.delenv("PARTLS_CHAIN_LEN", raising=False)
    ctx = partls.Context(0)                                          # the knobs are read once, at partls_create
    try:
        ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT if flags is None else flags)
        bo, bp, allo, unconv = ctx.opt_sweep(0, -1, want_all=True)
        pivots, vetoes = ctx.pivots(), ctx.vetoes()                  # of the sweep (the finish's node solve has counters of its own)
        a, b, t, opt, bi = ctx.opt_finish(bp)
        return dict(all=allo.copy(), bp=bp, bi=bi, opt=opt, alpha=a, unconv=unconv, pivots=pivots, vetoes=vetoes)
    finally:
        ctx.close()


def _oracle_sample(oracle, X, y, P, pats):
    Xo, Po = oracle.homogeneous(X, P)
    R, z = oracle.compress(Xo, y)
    return oracle.opt_patterns(R, z, Po, np.asarray(pats))


@pytest.mark.parametrize("D,K,chain_len", [(330, 7, None), (333, 8, None), (340, 10, 256), (352, 6, 16), (447, 6, None), (480, 5, None), (511, 5, None)])
def test_every_pattern_equals_the_eager_kernel_and_the_oracle(partls, oracle, monkeypatch, D, K, chain_len):
    """512-thread plan (n = D + 1 <= 512).  D = 330: the last wave with tableau rows ends below lane 16 (ld = 332: 12 rows) — the
    cross-lane reads of the kernel must not depend on lanes without a row; D = 447 / 480: the wave that runs phase 1 of the panel also
    owns 1 / 34 tableau rows; D = 511 (n = 512): the first size on the 1024-thread plan.  chain_len 256 at K = 10: eight chains of 256 patterns, ~150 flushes each; 16: chain starts
    dominate (the first pattern of a chain is solved from the empty basis: ~n/2 pivots in blocks of 16)."""
    X, y, P = _problem(1000 + D, 2 * D + 50, D, K)
    lz = _sweep(partls, monkeypatch, X, y, P, eager=False, chain_len=chain_len)
    eg = _sweep(partls, monkeypatch, X, y, P, eager=True, chain_len=chain_len)
    assert lz["unconv"] == 0 and eg["unconv"] == 0
    assert lz["bp"] == eg["bp"] == int(np.argmin(lz["all"]))
    np.testing.assert_allclose(lz["all"], eg["all"], rtol=1e-10)
    assert abs(lz["pivots"] - eg["pivots"]) <= 0.01 * eg["pivots"]          # the same walk (round-off may move a tie)
    pats = np.unique(np.concatenate([[lz["bi"]], np.random.default_rng(D).integers(0, 1 << (K + 1), 48)]))
    np.testing.assert_allclose(lz["all"][pats], _oracle_sample(oracle, X, y, P, pats), rtol=1e-9)
    np.testing.assert_allclose(lz["alpha"], eg["alpha"], atol=1e-9)


@pytest.mark.parametrize("D,K", [(520, 4), (700, 3)])
def test_beyond_511_variables(partls, oracle, monkeypatch, D, K):
    """1024-thread plan (fewer pending rows fit the LDS: flushes every ~15 pivots; the panel in the step-by-step form)"""
    X, y, P = _problem(2000 + D, 2 * D + 30, D, K)
    lz = _sweep(partls, monkeypatch, X, y, P, eager=False)
    eg = _sweep(partls, monkeypatch, X, y, P, eager=True)
    assert lz["unconv"] == 0 and lz["bp"] == eg["bp"]
    np.testing.assert_allclose(lz["all"], eg["all"], rtol=1e-10)
    pats = np.arange(1 << (K + 1))
    np.testing.assert_allclose(lz["all"], _oracle_sample(oracle, X, y, P, pats), rtol=1e-9)


@pytest.mark.parametrize("D,K,dup", [(336, 6, 8), (400, 5, 12)])
def test_exactly_dependent_columns(partls, oracle, monkeypatch, D, K, dup):
    """Copies and sums of columns (refused by the plain pivot test) and badly scaled dependent triples: entering pivots that only the
    leave-one-out rule refuses (sweep_blk.hip / gj_panel.h), i.e. the two-phase panel meets raised veto flags and redoes blocks with refused steps.  The oracle carries the same rule; objectives must agree."""
    X, y, P = _problem(3000 + D, 2 * D + 40, D, K, dup=dup, trip=10)
    lz = _sweep(partls, monkeypatch, X, y, P, eager=False)
    eg = _sweep(partls, monkeypatch, X, y, P, eager=True)
    assert lz["unconv"] == 0 and eg["unconv"] == 0
    assert lz["vetoes"] > 0                                              # the redo path of the panel ran
    np.testing.assert_allclose(lz["all"], eg["all"], rtol=1e-9)
    pats = np.unique(np.concatenate([[lz["bi"]], np.random.default_rng(D).integers(0, 1 << (K + 1), 24)]))
    np.testing.assert_allclose(lz["all"][pats], _oracle_sample(oracle, X, y, P, pats), rtol=1e-8)


def test_free_intercept_sharded_and_deterministic(partls, oracle, monkeypatch):
    """the benchmark mode (free intercept, 2^K patterns), three Gray-index shards == the full sweep, two runs bitwise equal"""
    D, K = 345, 9
    X, y, P = _problem(4001, 900, D, K)
    monkeypatch.delenv("PARTLS_EAGER_GENERIC", raising=False)
    ctx = partls.Context(0)
    try:
        ctx.opt_prepare(X, y, P, 0.0, 0)
        npat = ctx.num_patterns()
        assert npat == 1 << K
        bo, bp, _, unconv = ctx.opt_sweep(0, -1)
        piv = ctx.pivots()
        bo2, bp2, _, _ = ctx.opt_sweep(0, -1)
        assert unconv == 0 and bp == bp2 and bo == bo2 and ctx.pivots() == piv
        parts = [ctx.opt_sweep(*partls.dist.shard_range(npat, r, 3)) for r in range(3)]
        assert min((p[0], p[1]) for p in parts)[1] == bp
        a, b, t, opt, bi = ctx.opt_finish(bp)
    finally:
        ctx.close()
    ref = oracle.fit_opt(X, y, P)
    assert abs(opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
    np.testing.assert_allclose(a, ref["alpha"], atol=1e-7)


def test_bnb_node_batches_beyond_320(partls, oracle):
    """fit(BnB) at n = 331: node batches run in node mode on the deferred-update kernel (warm-started from the parent's snapshot since
    round 4); BnB optimum = Opt optimum = oracle (BnB.jl:94-132 explores the same sign patterns)."""
    X, y, P = _problem(5001, 800, 330, 4, noise=1.0)
    mb, _, rb = partls.fit(partls.BnB, X, y, P)
    mo, _, ro = partls.fit(partls.Opt, X, y, P)
    ref = oracle.fit_opt(X, y, P)
    assert abs(rb.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and abs(ro.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
    np.testing.assert_allclose(partls.predict(mb, X), partls.predict(mo, X), atol=1e-6 * np.linalg.norm(y))


def _branching_large(seed, N, D, K):
    """a target the partitioned model cannot explain: unconstrained signs inside every group, so the relaxation mixes signs and BnB branches"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    y = X @ (rng.standard_normal(D) * (rng.random(D) < 0.5)) + 0.5 * rng.standard_normal(N)
    return np.asfortranarray(X), y, np.asfortranarray(P)


@pytest.mark.parametrize("D,K", [(330, 6), (520, 4)])
def test_bnb_warm_starts_beyond_320_equal_cold_ones(partls, oracle, monkeypatch, D, K):
    """BnB node bounds beyond the register kernel start from the PARENT's tableau (a snapshot of the deferred-update kernel's state: base
    image with every pending term applied + rhs column + basis flags, BnB.jl:120-124) — same bounds, same branching, hence the same
    search (node for node) and the same optimum as with every node solved from the fresh tableau (PARTLS_BNB_COLD), in far fewer pivots.
    D = 330: ld mod 64 < 16 (the lane-mask corner of round 3); D = 520: the 1024-thread plan."""
    X, y, P = _branching_large(77, 900, D, K)
    res = {}
    for mode in ("warm", "cold"):
        if mode == "cold":
            monkeypatch.setenv("PARTLS_BNB_COLD", "1")
        monkeypatch.setenv("PARTLS_BNB_BATCH", "16")
        ctx = partls.Context(0)
        try:
            ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
            mu, pat, free, nodes = ctx.bnb_search(0)
            # one batch by hand: the children of the root, warm against cold, bound for bound
            ctx.bnb_snap_begin()
            lb0, br0, dst0 = ctx.bnb_bound_snap(np.array([0], dtype=np.uint64), np.array([(1 << (K + 1)) - 1], dtype=np.uint64), np.array([-1], dtype=np.int32))
            piv_root = ctx.pivots()
            k = int(br0[0]); bit = 1 << k; fr = ((1 << (K + 1)) - 1) & ~bit
            lb1, br1, dst1 = ctx.bnb_bound_snap(np.array([bit, 0], dtype=np.uint64), np.array([fr, fr], dtype=np.uint64), np.array([dst0[0], dst0[0]], dtype=np.int32))
            piv_children = ctx.pivots()
            a, b, t, opt = ctx.bnb_leaf(pat, free)
            res[mode] = dict(mu=mu, pat=pat, free=free, nodes=nodes, lb1=lb1.copy(), br1=br1.copy(), dst0=int(dst0[0]), piv_root=piv_root, piv_children=piv_children, opt=opt)
        finally:
            ctx.close()
    w, c = res["warm"], res["cold"]
    assert w["nodes"] > 8 and k >= 0
    assert w["dst0"] >= 0 and c["dst0"] == -1                                  # snapshots are taken beyond n = 320 now; not in cold mode
    assert (w["pat"], w["free"], w["nodes"]) == (c["pat"], c["free"], c["nodes"])
    assert abs(w["mu"] - c["mu"]) <= 1e-10 * c["mu"] and abs(w["opt"] - c["opt"]) <= 1e-10 * c["opt"]
    np.testing.assert_allclose(w["lb1"], c["lb1"], rtol=1e-10)
    np.testing.assert_array_equal(w["br1"], c["br1"])
    assert w["piv_children"] * 4 < c["piv_children"], (w["piv_children"], c["piv_children"])      # a child exchanges one group, not half the variables
    ref = oracle.fit_bnb(X, y, P)
    assert abs(w["opt"] - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


def test_a_lost_progress_word_ends_in_not_converged_not_in_a_hang(partls, monkeypatch):
    """The two-phase panel (n >= 512 since round 4; below, the panel takes one barrier per step and polls nothing): its followers wait for
    the progress word of the wave that runs phase 1 — a bounded wait (2^22 polls).  Fault
    injection (PARTLS_LZ_FAULT): workgroup 0's first panel never publishes the word.  The kernel must finish (every wave reaches the bound
    and walks on), and the sweep must not pass the garbage off as a result: the unconverged count is raised, fit() raises status 6."""
    import time
    X, y, P = _problem(31, 900, 520, 3)                              # n >= 512: the 1024-thread plan, where the two-phase panel runs
    monkeypatch.setenv("PARTLS_LZ_FAULT", "1")
    ctx = partls.Context(0)
    try:
        ctx.opt_prepare(X, y, P, 0.0, 0)
        t0 = time.time()
        bo, bp, _, unconv = ctx.opt_sweep(0, -1)
        assert unconv >= 1 and time.time() - t0 < 60.0
    finally:
        ctx.close()
    monkeypatch.delenv("PARTLS_LZ_FAULT")
    ctx = partls.Context(0)
    try:
        ctx.opt_prepare(X, y, P, 0.0, 0)
        assert ctx.opt_sweep(0, -1)[3] == 0
    finally:
        ctx.close()
