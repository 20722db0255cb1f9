"""CPU baselines of bench.py (TEST / MEASUREMENT INFRASTRUCTURE — only bench.py's cpu_baseline leg imports this module).

Baseline A ("port", reference-faithful): the loop body of fit(Opt), Opt.jl:87-90, per sampled pattern — column scaling, dense
Lawson–Hanson NNLS on the N x (D+1) matrix, objective — by the C oracle, one pattern per call, spread over worker processes
(the reference itself is single-threaded Julia; the per-core rate is reported next to the aggregate).
Baseline B (the fair algorithmic comparison): the same NNLS on the QR-compressed problem [R z] (the CPU
analogue of working from the Gram block: X is read once), on 1 core and on all worker processes.
Workers are spawned (never forked from a process that has initialised the GPU) and rebuild the synthetic inputs themselves.
"""
import multiprocessing as mp
import os
import time

import numpy as np

_G = {}


def _init(seed, N, D, K, compressed):
    from oracle import oracle as O
    X, y, P, _ = O.synth(seed, N, D, K)
    Xo, Po = O.homogeneous(X, P)
    if compressed:
        R, z = O.compress(Xo, y)
        _G.update(A=R, b=z, Po=Po)
    else:
        _G.update(A=Xo, b=y, Po=Po)


def _solve(pats):
    from oracle import oracle as O
    t0 = time.perf_counter()
    objs = O.opt_patterns(_G["A"], _G["b"], _G["Po"], np.asarray(pats, dtype=np.int64))
    return list(map(float, objs)), time.perf_counter() - t0


def _run(seed, N, D, K, compressed, chunks, nproc):
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(nproc, initializer=_init, initargs=(seed, N, D, K, compressed)) as pool:
        pool.map(_solve, [[0]] * nproc)                      # every worker has built its inputs (and is warm)
        t1 = time.perf_counter()
        res = pool.map(_solve, chunks, chunksize=1)
        t2 = time.perf_counter()
    objs = [o for r in res for o in r[0]]
    return objs, t2 - t1, sum(r[1] for r in res), t1 - t0


def dense_sample(seed, N, D, K, patterns, nproc):
    """Baseline A on `patterns` with nproc workers: objectives, wall seconds of the solves, summed CPU seconds, setup seconds."""
    return _run(seed, N, D, K, False, [[int(p)] for p in patterns], nproc)


def compressed_rates(seed, N, D, K, patterns_per_worker, nproc):
    """Baseline B: solves/s on 1 core and on nproc cores (compression done once per worker beforehand, not timed)."""
    rng = np.random.default_rng(1)
    npat = 1 << (K + 1)
    one = [list(map(int, rng.integers(0, npat, patterns_per_worker)))]
    _, wall1, _, setup1 = _run(seed, N, D, K, True, one, 1)
    many = [list(map(int, rng.integers(0, npat, patterns_per_worker))) for _ in range(nproc)]
    _, walln, _, _ = _run(seed, N, D, K, True, many, nproc)
    return patterns_per_worker / wall1, patterns_per_worker * nproc / walln, setup1


def default_workers():
    return max(1, min(16, (os.cpu_count() or 1)))
