"""GPU tests that put every BASELINE config and every size limit of the kernels against the oracle (round-2 additions):
C4 at its full N, Alt at D = 512 from the oracle's start, the global-memory tableau at M = 700 / 1021, overlapping partitions
for Opt / Alt / BnB, exact ties, a BnB instance that really branches at D = 256, device-resident predict, and the 2-rank
control flow of bench.py on one device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _device_problem(partls, seed, N, D, K):
    import torch
    P, ws = partls.synth_truth(seed, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device="cuda")
    dy = torch.empty(N, dtype=torch.float64, device="cuda")
    ctx = partls.default_context()
    ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    return ctx, dX, dy, P, ws


def test_c4_full_size_alt(partls):
    """BASELINE config 4 at its stated size (N = 1 000 000, D = 512, K = 16; 4.1 GB of X generated in HBM): Alt from a random
    start on the n = 513 tableau (cooperative global-memory kernel) reaches the noise floor; every iterate is a valid model;
    the objective from the Gram equals the objective from the data (Alt.jl:112-113)."""
    import torch
    seed, N, D, K = 20260004, 1_000_000, 512, 16
    ctx, dX, dy, P, ws = _device_problem(partls, seed, N, D, K)
    ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    rng = np.random.default_rng(123)
    a0 = rng.random(D + 1); b0 = (rng.random(K + 1) - 0.5) * 10
    a, b, t, opt, iters = ctx.alt_prepared(a0, b0, eps=1e-6, T=200)
    assert 1 <= iters <= 200 and np.all(a >= 0)
    grp = np.argmax(P, axis=1)
    for k in range(K):
        assert abs(a[grp == k].sum() - 1.0) < 1e-9            # alpha normalised per group (Alt.jl:95-98)
    G = ctx.gram()
    w = np.concatenate([a * b[grp], [t]])
    obj = np.sqrt(w @ G[:D + 1, :D + 1] @ w - 2 * w @ G[:D + 1, D + 1] + G[D + 1, D + 1])
    assert abs(obj - opt) <= 1e-8 * opt
    assert opt < 1.5 * 0.1 * np.sqrt(N)                       # Alt is a local method; it gets close to the noise floor here
    # the Gram itself against a float64 torch reference on a column sample (full X'X would be another 526 GFLOP)
    X = dX.view(D, N)                                          # column-major N x D == row-major D x N
    cols = [0, 1, 255, 256, 510, 511]
    ref = (X[cols] @ X.T).cpu().numpy()                        # 6 x D
    np.testing.assert_allclose(G[cols, :D], ref, rtol=1e-11, atol=1e-6)
    del dX, dy
    torch.cuda.empty_cache()


def test_alt_d512_vs_oracle_same_start(partls, oracle):
    """Alt at the C4 feature shape (D = 512, K = 16 -> n = 513 > 320: cooperative kernel for the alpha-step) on an N the dense
    oracle can afford, from the same (alpha0, beta0): Alt.jl:77-117 iterates are deterministic, so objective and model agree."""
    seed, N, D, K = 20260004, 2000, 512, 16
    X, y, P, _ = oracle.synth(seed, N, D, K)
    rng = np.random.default_rng(5)
    a0 = rng.random(D + 1); b0 = (rng.random(K + 1) - 0.5) * 10
    T = 3
    ref = oracle.fit_alt(X, y, P, a0, b0, eta=0.0, eps=1e-12, T=T)
    m, _, rep = partls.fit(partls.Alt, X, y, P, η=0.0, ϵ=1e-12, T=T, alpha0=a0, beta0=b0)
    assert rep.iters == ref["iters"] == T
    assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(partls.predict(m, X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]),
                               atol=1e-6 * max(1.0, float(np.linalg.norm(y))))


@pytest.mark.parametrize("D", [700, 1021])
def test_global_memory_tableau_at_its_size_limit(partls, oracle, D):
    """M = 700 and the documented maximum M + 1 = 1022 (check_common): every pattern of fit(Opt) on the global-memory kernels
    (sweep_generic chains, sweep_coop winner re-solve) against the oracle; Alt from the same start."""
    rng = np.random.default_rng(D)
    K = 2
    N = 2 * D + 50
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), rng.integers(0, K, size=D)] = 1
    X = rng.standard_normal((N, D))
    grp = P.argmax(1)
    y = X @ (rng.random(D) * np.array([1.5, -0.7])[grp]) + 0.3 + 0.1 * rng.standard_normal(N)
    ref = oracle.fit_opt(X, y, P, return_all=True)
    model, _, rep = partls.fit(partls.Opt, X, y, P, returnAllSolutions=True)
    got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
    np.testing.assert_allclose(got, ref["all_opt"], rtol=1e-8)
    m2, _, r2 = partls.fit(partls.Opt, X, y, P)
    assert abs(r2.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and r2.best_index == ref["best_index"]
    a0 = np.random.default_rng(3).random(D + 1); b0 = (np.random.default_rng(4).random(K + 1) - 0.5) * 10
    ra = oracle.fit_alt(X, y, P, a0, b0, eps=1e-9, T=2)
    m3, _, r3 = partls.fit(partls.Alt, X, y, P, ϵ=1e-9, T=2, alpha0=a0, beta0=b0)
    assert abs(r3.opt - ra["opt"]) <= 1e-8 * max(1.0, ra["opt"])


def _overlap_problem(seed=11, N=100):
    """the partition of the docstring example at Opt.jl:68 (features 4 and 5 sit in two groups each)"""
    rng = np.random.default_rng(seed)
    P = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1]], dtype=np.int64)
    X = rng.random((N, 5))
    y = rng.random(N)
    return X, y, P


def test_opt_overlapping_partition_vs_oracle(partls, oracle):
    """Overlapping P (Opt.jl:68): bmatrix multipliers f = sum_k P[m,k] s_k in {0, ±1, ±2} (Opt.jl:28-29); every pattern's optval
    and the cleaned-up winner (Opt.jl:34-44) against the oracle's literal restatement."""
    for seed, eta in ((11, 0.0), (12, 0.3)):
        X, y, P = _overlap_problem(seed)
        ref = oracle.fit_opt(X, y, P, eta=eta, return_all=True)
        model, _, rep = partls.fit(partls.Opt, X, y, P, η=eta, returnAllSolutions=True)
        got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
        np.testing.assert_allclose(got, ref["all_opt"], rtol=1e-9, atol=1e-12)
        m2, _, r2 = partls.fit(partls.Opt, X, y, P, η=eta, faithful_intercept=True)
        assert abs(r2.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
        if r2.best_index == ref["best_index"]:
            np.testing.assert_allclose(m2.α, ref["alpha"], atol=1e-7)
            np.testing.assert_allclose(m2.β, ref["beta"], atol=1e-7)
            assert abs(m2.t - ref["t"]) < 1e-7


def test_alt_and_bnb_overlapping_partition_vs_oracle(partls, oracle):
    """The reference accepts any 0/1 P (PartitionedLS.jl:292): Alt's multipliers sum_k P[m,k] beta_k (Alt.jl:80-81) and BnB's
    accumulated per-feature constraints (BnB.jl:120-121) on the device, against the oracle."""
    X, y, P = _overlap_problem(21, N=80)
    M, K = P.shape
    rng = np.random.default_rng(1)
    for T in (1, 4):
        a0 = rng.random(M + 1); b0 = (rng.random(K + 1) - 0.5) * 10
        ref = oracle.fit_alt(X, y, P, a0, b0, eta=0.0, eps=1e-12, T=T)
        m, _, rep = partls.fit(partls.Alt, X, y, P, η=0.0, ϵ=1e-12, T=T, alpha0=a0, beta0=b0)
        assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
        np.testing.assert_allclose(m.α, ref["alpha"], atol=1e-7)
        np.testing.assert_allclose(m.β, ref["beta"], atol=1e-7)
    for seed in (31, 32, 33):
        X, y, P = _overlap_problem(seed, N=60)
        y = y - y.mean() + 0.2 * X[:, 0] - 0.4 * X[:, 3]        # mixed signs inside the overlapping groups: the search branches
        ref = oracle.fit_bnb(X, y, P)
        m, _, rep = partls.fit(partls.BnB, X, y, P)
        assert abs(rep.opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
        np.testing.assert_allclose(partls.predict(m, X), oracle.predict(X, P, ref["alpha"], ref["beta"], ref["t"]), atol=1e-6)


def test_exact_tie_returns_first_index(partls, oracle):
    """A group without features makes the two patterns that differ in its sign the same subproblem; argmin returns the first
    index (Opt.jl:96), i.e. the one with that group's bit clear — in both intercept modes and for any position of the group."""
    rng = np.random.default_rng(3)
    N, M = 200, 9
    X = rng.standard_normal((N, M))
    for empty in (0, 1, 3):
        K = 4
        P = np.zeros((M, K), dtype=np.int64)
        others = [k for k in range(K) if k != empty]
        P[np.arange(M), np.array(others)[np.arange(M) % 3]] = 1
        w = np.array([1.0, -2.0, 0.5])[np.arange(M) % 3] * rng.random(M)
        y = X @ w + 0.7 + 0.05 * rng.standard_normal(N)
        ref = oracle.fit_opt(X, y, P, return_all=True)
        assert ref["all_opt"][ref["best_index"]] == ref["all_opt"][ref["best_index"] ^ (1 << empty)]   # the oracle sees the tie
        assert not (ref["best_index"] >> empty) & 1
        for faithful in (False, True):
            m, _, rep = partls.fit(partls.Opt, X, y, P, faithful_intercept=faithful)
            assert rep.best_index == ref["best_index"], (empty, faithful, rep.best_index, ref["best_index"])
            assert abs(rep.opt - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


def test_bnb_branches_at_d256_and_equals_opt(partls):
    """fit(BnB) at the C5 feature shape with a target the model cannot explain (pure noise): the root relaxation is far from
    feasible, so the best-first frontier really runs in device batches of 512 nodes (BnB.jl:99-124); optimum == Opt's."""
    rng = np.random.default_rng(7)
    N, D, K = 3000, 256, 14
    X = rng.standard_normal((N, D))
    y = rng.standard_normal(N)
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), np.arange(D) % K] = 1
    m1, _, r1 = partls.fit(partls.Opt, X, y, P)
    m2, _, r2 = partls.fit(partls.BnB, X, y, P)
    assert r2.nopen > 512, r2.nopen
    assert abs(r1.opt - r2.opt) <= 1e-9 * r1.opt
    np.testing.assert_allclose(partls.predict(m1, X), partls.predict(m2, X), atol=1e-7)


def test_predict_device_pointers(partls, oracle):
    """predict with X and yhat resident in HBM (partls_predict_device): same numbers as the host-pointer entry and the oracle."""
    import torch
    X, y, P, _ = oracle.synth(20260001, 5000, 40, 5)
    model, _, rep = partls.fit(partls.Opt, X, y, P)
    dX = torch.tensor(np.asfortranarray(X).T.copy(), dtype=torch.float64, device="cuda")      # D x N row-major == N x D column-major
    dyh = torch.zeros(X.shape[0], dtype=torch.float64, device="cuda")
    partls.predict_device(model, dX.data_ptr(), X.shape[0], X.shape[0], dyh.data_ptr())
    torch.cuda.synchronize()
    host = partls.predict(model, X)
    np.testing.assert_allclose(dyh.cpu().numpy(), host, rtol=0, atol=1e-12)
    np.testing.assert_allclose(host, oracle.predict(X, P, model.α, model.β, model.t), atol=1e-9)
    # a strided view: leading dimension larger than N
    big = torch.zeros((40, 5100), dtype=torch.float64, device="cuda")
    big[:, :5000] = dX
    partls.predict_device(model, big.data_ptr(), 5000, 5100, dyh.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(dyh.cpu().numpy(), host, rtol=0, atol=1e-12)


def test_bench_two_rank_control_flow_on_one_device(partls):
    """bench.py's N = 2 path (Gray-index shards, all-reduce(min objective) then min index, winner re-solve on every rank) with
    both ranks on device 0 over gloo: same winner and objective as the 1-rank run."""
    env = dict(os.environ, PARTLS_BENCH_SHARE_GPU="1", PARTLS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C2",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j1["result"]["best_index"] == j2["result"]["best_index"]
    assert abs(j1["result"]["opt"] - j2["result"]["opt"]) <= 1e-12 * j1["result"]["opt"]


def _alt_problem_n513(oracle, seed):
    """D = 512 features -> n = 513 tableau variables: every alpha-step runs on the multi-workgroup (grid barrier) kernel"""
    N, D, K = 1500, 512, 8
    X, y, P, _ = oracle.synth(seed, N, D, K)
    rng = np.random.default_rng(seed)
    return X, y, P, rng.random(D + 1), (rng.random(K + 1) - 0.5) * 10


def test_two_contexts_run_the_grid_barrier_kernel_concurrently(partls, oracle):
    """Two contexts on two host threads run Alt at n = 513 at the same time: two grids with hand-written grid barriers share the
    device.  Both must finish with the oracle's result (a barrier that cannot complete times out after 2 s of wall clock and the
    solve is repeated on the one-workgroup kernel) — never a hang."""
    import threading
    probs = [_alt_problem_n513(oracle, 20260041 + i) for i in range(2)]
    refs = [oracle.fit_alt(X, y, P, a0, b0, T=6) for X, y, P, a0, b0 in probs]
    ctxs = [partls.Context(0) for _ in probs]
    out = [None, None]

    def run(i):
        X, y, P, a0, b0 = probs[i]
        try:
            for _ in range(3):                                            # several fits each, so that the launches really interleave
                ctxs[i].opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
                out[i] = ctxs[i].alt_prepared(a0, b0, eps=1e-6, T=6)
        except Exception as e:                                            # noqa: BLE001
            out[i] = e

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a grid-barrier kernel hung"
    for i in range(2):
        assert not isinstance(out[i], Exception), out[i]
        a, b, t, opt, iters = out[i]
        assert abs(opt - refs[i]["opt"]) <= 1e-8 * max(1.0, refs[i]["opt"])
        np.testing.assert_allclose(a, refs[i]["alpha"], atol=1e-6)
    for c in ctxs:
        c.close()


def test_grid_barrier_timeout_falls_back_to_one_workgroup(partls, oracle, monkeypatch):
    """PARTLS_COOP_FAULT makes every grid barrier of the cooperative kernel wait for one arrival too many, i.e. behave as if part of
    the grid were not resident: the kernel must abort within its 2 s wall-clock bound (abort word: all workgroups leave together)
    and the solve must be repeated on the one-workgroup kernel with the right answer."""
    import time
    X, y, P, a0, b0 = _alt_problem_n513(oracle, 20260043)
    ref = oracle.fit_alt(X, y, P, a0, b0, T=2)
    monkeypatch.setenv("PARTLS_COOP_FAULT", "1")
    ctx = partls.Context(0)
    ctx.opt_prepare(X, y, P, 0.0, partls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    t0 = time.perf_counter()
    a, b, t, opt, iters = ctx.alt_prepared(a0, b0, eps=1e-6, T=2)
    dt = time.perf_counter() - t0
    ctx.close()
    assert abs(opt - ref["opt"]) <= 1e-8 * max(1.0, ref["opt"])
    np.testing.assert_allclose(a, ref["alpha"], atol=1e-6)
    assert 2.0 * iters <= dt < 4.0 * iters + 5.0, dt                      # one ~2 s timeout per alpha-step, not one per workgroup


def _bnb_warm_gpu_worker(rank, world, port, out):
    import numpy as np
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(17)
    N, D, K = 500, 48, 8
    X = rng.standard_normal((N, D)); y = X @ rng.standard_normal(D) + 0.2 * rng.standard_normal(N)
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    ctx = pls.Context(0)                                         # both ranks share device 0: each has its own snapshot pool
    ctx.opt_prepare(X, y, P, 0.0, pls.lowlevel.OPT_FAITHFUL_INTERCEPT)
    res = pls.dist.bnb_search_warm(ctx, K + 1, rank=rank, world=world, batch=16)
    a, b, t, opt = ctx.bnb_leaf(res[1], res[2])
    out[rank] = (res, opt)
    ctx.close()
    dist.destroy_process_group()


def test_bnb_search_warm_two_ranks_on_one_device(partls, oracle):
    """dist.bnb_search_warm with 2 gloo ranks that share device 0 (each with its own context and snapshot pool): nodes are dealt to the
    rank that holds the parent's snapshot, surplus goes cold to the other — same optimum and node count on both ranks, same as the
    in-library search and the oracle."""
    import socket
    import torch.multiprocessing as mp
    rng = np.random.default_rng(17)
    N, D, K = 500, 48, 8
    X = rng.standard_normal((N, D)); y = X @ rng.standard_normal(D) + 0.2 * rng.standard_normal(N)
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    ref = oracle.fit_bnb(X, y, P)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctxm = mp.get_context("spawn")
    out = ctxm.Manager().dict()
    procs = [ctxm.Process(target=_bnb_warm_gpu_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    (mu0, pat0, free0, n0), opt0 = out[0]
    (mu1, pat1, free1, n1), opt1 = out[1]
    assert (pat0, free0, n0) == (pat1, free1, n1) and mu0 == mu1 and n0 > 30
    assert abs(opt0 - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and abs(mu0 - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])


def test_staged_upload_is_bit_identical_to_the_device_resident_path(partls, oracle, monkeypatch):
    """A host X larger than 8 MB goes up through page-locked staging buffers (4 copier threads, column ranges): the device image, hence the
    Gram products and everything after them, must be the one the plain copy and the device-resident path produce — with a leading
    dimension, with columns longer than a staging buffer, and through predict."""
    import torch
    rng = np.random.default_rng(5)
    for N, D, K, ld in ((70_000, 24, 4, 70_000), (60_001, 30, 5, 60_017), (1_200_000, 3, 2, 1_200_000)):
        Xbig = np.asfortranarray(rng.standard_normal((ld, D)))
        X = Xbig[:N]                                                            # F-ordered view with leading dimension ld
        P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
        y = X @ (rng.random(D) * (np.arange(D) % K - 1.5)) + 0.1 * rng.standard_normal(N)
        L = partls.lowlevel
        ctx = partls.Context(0)
        lib = L.lib()
        import ctypes as C
        Pf = np.asfortranarray(P)
        def prep(flag_env):
            assert lib.partls_opt_prepare(ctx._h, Xbig.ctypes.data, N, D, ld, y.ctypes.data, 0, Pf.ctypes.data, K, D, 0.0, 0) == L.OK
            ctx._shape = (N, D, K)
            return ctx.gram().copy(), ctx.upload()
        G1, up1 = prep(None)
        assert up1[1] == N * D * 8 and up1[0] > 0
        dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()                 # column-major N x D on the device
        dy = torch.from_numpy(y).cuda()
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
        G2 = ctx.gram().copy()
        assert ctx.upload() == (0.0, 0.0)
        np.testing.assert_array_equal(G1, G2)
        ctx.close()
        monkeypatch.setenv("PARTLS_NO_STAGED_UPLOAD", "1")
        ctx = partls.Context(0)
        G3, _ = prep(None)
        np.testing.assert_array_equal(G1, G3)
        ctx.close()
        monkeypatch.delenv("PARTLS_NO_STAGED_UPLOAD")
    # predict through the staged upload
    X = np.asfortranarray(rng.standard_normal((200_000, 8)))
    P = np.zeros((8, 2), dtype=np.int64); P[:4, 0] = 1; P[4:, 1] = 1
    model = partls.PartLSFitResult(rng.random(8), np.array([1.5, -2.0]), 0.25, P)
    np.testing.assert_allclose(partls.predict(model, X), X @ (model.α * model.β[np.argmax(P, axis=1)]) + 0.25, rtol=0, atol=1e-12)
