import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, partls_amd
from oracle import oracle as O
pls = partls_amd.package(); O.build()
for D in (700, 1021):
    rng = np.random.default_rng(D); K = 2; N = 2 * D + 50
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), rng.integers(0, K, size=D)] = 1
    X = rng.standard_normal((N, D)); grp = P.argmax(1)
    y = X @ (rng.random(D) * np.array([1.5, -0.7])[grp]) + 0.3 + 0.1 * rng.standard_normal(N)
    t = time.time(); ref = O.fit_opt(X, y, P, return_all=True); to = time.time() - t
    t = time.time(); model, _, rep = pls.fit(pls.Opt, X, y, P, returnAllSolutions=True); tg = time.time() - t
    got = np.array([rep.solutions._all[b] for b in range(len(ref["all_opt"]))])
    print(f"D={D}: oracle {to:.1f} s, gpu {tg*1e3:.1f} ms, max rel diff {np.max(np.abs(got-ref['all_opt'])/np.maximum(1,ref['all_opt'])):.2e}")
    m2, _, r2 = pls.fit(pls.Alt, X, y, P, T=5, rng=3)
    a0 = np.random.default_rng(3).random(D + 1); b0 = (np.random.default_rng(4).random(K + 1) - 0.5) * 10
    ra = O.fit_alt(X, y, P, a0, b0, eps=1e-9, T=4); m3, _, r3 = pls.fit(pls.Alt, X, y, P, ϵ=1e-9, T=4, alpha0=a0, beta0=b0)
    print(f"   Alt same start: oracle {ra['opt']:.10f} gpu {r3.opt:.10f}")
