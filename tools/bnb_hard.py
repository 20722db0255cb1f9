import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, partls_amd
pls = partls_amd.package()
for (N, D, K, seed) in ((2000, 120, 12, 1), (3000, 200, 16, 2)):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D)); y = rng.standard_normal(N)            # no signal at all
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), np.arange(D) % K] = 1
    t = time.time(); m1, _, r1 = pls.fit(pls.Opt, X, y, P); t1 = time.time() - t
    t = time.time(); m2, _, r2 = pls.fit(pls.BnB, X, y, P); t2 = time.time() - t
    print(f"D={D} K={K}: Opt {r1.opt:.10f} in {t1*1e3:.1f} ms | BnB {r2.opt:.10f} in {t2*1e3:.1f} ms, nopen {getattr(r2, 'nopen', None)} | gap {abs(r1.opt-r2.opt):.2e}")
