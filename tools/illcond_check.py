import sys; sys.path.insert(0,'/root/repo')
import numpy as np
import partls_amd
from oracle import oracle as O
pls = partls_amd.package()
rng = np.random.default_rng(42)
for noise in [1e-2, 1e-3, 1e-4, 1e-5]:
    N, D, K = 2000, 24, 4
    Z = rng.standard_normal((N, 6))
    X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))     # cond(X) ~ 1/noise
    grp = np.arange(D) % K
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
    y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
    ref = O.fit_opt(X, y, P)
    m, _, rep = pls.fit(pls.Opt, X, y, P)
    sv = np.linalg.svd(np.hstack([X, np.ones((N,1))]), compute_uv=False)
    print(f"noise={noise:g} cond(Xo)={sv[0]/sv[-1]:.2e} opt gpu={rep.opt:.12f} ref={ref['opt']:.12f} gap={abs(rep.opt-ref['opt'])/max(1,ref['opt']):.2e} "
          f"best {rep.best_index}/{ref['best_index']} max|dalpha|={np.abs(m.α-ref['alpha']).max():.2e} max|dbeta|={np.abs(m.β-ref['beta']).max():.2e} |dt|={abs(m.t-ref['t']):.2e}")
