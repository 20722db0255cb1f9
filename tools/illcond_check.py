"""illcond_check.py — what fit(Opt) / fit(BnB) / fit(Alt) return as the data get ill-conditioned (cond(Xo) ~ 1/noise): objective / model gap to the dense
oracle, the data-space KKT violation of the winner (partls_get_kkt_violation), leave-one-out vetoes of the sweep, and the status."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import partls_amd
from oracle import oracle as O
pls = partls_amd.package()


def problem(noise, seed=42, N=2000, D=24, K=4):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((N, 6))
    X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))     # cond(X) ~ 1/noise
    grp = np.arange(D) % K
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
    y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
    return X, y, P


if __name__ == "__main__":
    for seed in (42, 43, 44):
        for noise in [1e-2, 1e-3, 1e-4, 1e-5, 3e-6, 1e-6, 3e-7, 1e-7]:
            X, y, P = problem(noise, seed)
            ref = O.fit_opt(X, y, P)
            ctx = pls.default_context()
            sv = np.linalg.svd(np.hstack([X, np.ones((X.shape[0], 1))]), compute_uv=False)
            for alg in (pls.Opt, pls.BnB):
                status = "OK"
                try:
                    m, _, rep = pls.fit(alg, X, y, P, on_ill_conditioned="raise")
                    opt, a = rep.opt, m.α
                except pls.PartlsError as e:
                    status, opt, a = f"status {e.status}", float("nan"), np.full(X.shape[1], np.nan)
                print(f"seed={seed} noise={noise:g} cond(Xo)={sv[0]/sv[-1]:.2e} {alg.__name__} {status:9s} kkt={ctx.kkt_violation():.2e} min_pivot={ctx.min_pivot():.1e} "
                      f"vetoes={ctx.vetoes()} gap={(opt-ref['opt'])/max(1,ref['opt']):+.2e} max|dalpha|={np.abs(a-ref['alpha']).max():.2e}")
            # Alt from a fixed start against the oracle's dense Alt from the same start (Alt.jl:77-117: QR beta-step, dense NNLS alpha-step)
            rng = np.random.default_rng(seed + 1000)
            a0, b0 = rng.random(X.shape[1] + 1), (rng.random(P.shape[1] + 1) - 0.5) * 10
            refa = O.fit_alt(X, y, P, a0, b0, eps=1e-9, T=60)
            m, _, rep = pls.fit(pls.Alt, X, y, P, alpha0=a0, beta0=b0, eps=1e-9, T=60)          # default: warn, model returned
            status = "status 9" if rep.get("ill_conditioned") else "OK"
            print(f"seed={seed} noise={noise:g} cond(Xo)={sv[0]/sv[-1]:.2e} Alt {status:9s} kkt={ctx.kkt_violation():.2e} iters={rep.iters} "
                  f"gap={(rep.opt-refa['opt'])/max(1,refa['opt']):+.2e} max|dalpha|={np.abs(m.α-refa['alpha']).max():.2e}")
