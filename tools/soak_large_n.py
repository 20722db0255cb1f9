#!/usr/bin/env python3
"""Randomised soak of the global-memory kernels (n > 320: sweep_generic enumeration, sweep_coop single solves) against the oracle:
fit(Alt) from the same start and fit(Opt) on problems with 321..420 features, some with dependent columns.
usage: python tools/soak_large_n.py [problems] [seed] [Dmin Dmax]   (default 321..420; 257..320 exercises the register kernel at T = 17..20)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, partls_amd
from oracle import oracle as O
pls = partls_amd.package()
nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
dlo = int(sys.argv[3]) if len(sys.argv) > 4 else 321
dhi = int(sys.argv[4]) if len(sys.argv) > 4 else 420
bad = 0
for it in range(nprob):
    D = int(rng.integers(dlo, dhi + 1)); K = int(rng.integers(2, 5)); N = int(D * rng.uniform(1.3, 2.5)) + 7
    X = rng.standard_normal((N, D))
    kind = rng.random()
    if kind < 0.3:
        i, j = rng.choice(D, 2, replace=False); X[:, i] = X[:, j] * rng.uniform(0.5, 2.0)          # exactly dependent pair
    elif kind < 0.5:
        i, j, l = rng.choice(D, 3, replace=False); X[:, i] = 0.5 * X[:, j] - 2.0 * X[:, l]
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), rng.integers(0, K, size=D)] = 1
    grp = P.argmax(1)
    y = X @ (rng.random(D) * ((rng.random(K) - 0.5) * 6)[grp]) + rng.uniform(-2, 2) + rng.choice([1e-3, 0.3, 3.0]) * rng.standard_normal(N)
    tag = f"problem {it}: N={N} D={D} K={K} kind={kind:.2f}"
    a0 = rng.random(D + 1); b0 = (rng.random(K + 1) - 0.5) * 10
    ra = O.fit_alt(X, y, P, a0, b0)
    m, _, rep = pls.fit(pls.Alt, X, y, P, alpha0=a0, beta0=b0)
    ok_alt = abs(rep.opt - ra["opt"]) <= 1e-7 * max(1.0, ra["opt"])
    ro = O.fit_opt(X, y, P)
    mo, _, repo = pls.fit(pls.Opt, X, y, P)
    ok_opt = abs(repo.opt - ro["opt"]) <= 1e-8 * max(1.0, ro["opt"])
    yh = pls.predict(mo, X); yr = O.predict(X, P, ro["alpha"], ro["beta"], ro["t"])
    ok_fit = np.linalg.norm(yh - yr) <= 1e-6 * max(1.0, float(np.linalg.norm(y)))
    if not (ok_alt and ok_opt and ok_fit):
        bad += 1
        print("MISMATCH", tag, "alt", rep.opt, ra["opt"], "opt", repo.opt, ro["opt"], "fit", float(np.linalg.norm(yh - yr)), flush=True)
    elif it % 5 == 0:
        print("ok", tag, flush=True)
print(f"{nprob - bad} of {nprob} problems agree with the oracle")
sys.exit(1 if bad else 0)
