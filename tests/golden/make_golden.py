#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run once, in the build container; outputs are committed).

The reference (/root/reference, pure Julia) cannot run in this image (no Julia), and its NNLS arithmetic lives in
the absent third-party package NonNegLeastSquares.jl.  These fixtures are therefore produced by an INDEPENDENT
numpy restatement of the reference flow (Opt.jl:73-104, Alt.jl:50-124, BnB.jl:30-132) that uses
scipy.optimize.nnls (Lawson–Hanson, scipy 1.15.3) as the NNLS solver — a different code base from oracle/partls_oracle.c,
so oracle-vs-golden agreement is a genuine cross-check.  The toy case additionally carries the reference's own
known answers (test/runtests.jl:9-36: opt ≈ 0, predictions == y; exact rationals from SURVEY.md §8c).

Fixtures are data only: inputs + expected outputs, as .npz.
"""
import os
import sys

import numpy as np
from scipy.optimize import nnls

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # only for the synthetic-input generator (inputs are stored in the fixture anyway)


def homogeneous(X, P):
    N, M = X.shape
    K = P.shape[1]
    Xo = np.hstack([X, np.ones((N, 1))])
    Po = np.zeros((M + 1, K + 1), dtype=np.int64)
    Po[:M, :K] = P
    Po[M, K] = 1
    return Xo, Po


def regularize(Xo, y, Po, eta):
    if eta == 0:
        return Xo, y
    rows = [np.sqrt(eta) * (Po[:, k] == 1).astype(float) for k in range(Po.shape[1])]
    return np.vstack([Xo] + [r[None, :] for r in rows]), np.concatenate([y, np.zeros(Po.shape[1])])


def ref_opt(X, y, P, eta=0.0):
    Xo, Po = homogeneous(X, P)
    Xo, yo = regularize(Xo, y, Po, eta)
    Kp = Po.shape[1]
    res = []
    for b in range(2 ** Kp):
        beta = np.array([2 * ((b >> k) & 1) - 1 for k in range(Kp)], dtype=float)
        f = (Po * beta[None, :]).sum(axis=1)
        a, _ = nnls(Xo * f[None, :], yo, maxiter=30 * Xo.shape[1])
        optval = np.linalg.norm(Xo @ ((Po * a[:, None]) @ beta) - yo)
        res.append((optval, a[:-1].copy(), beta[:-1].copy(), beta[-1] * a[-1]))
    objs = np.array([r[0] for r in res])
    bi = int(np.argmin(objs))

    def cleanup(r):
        _, a, b, t = r
        A = (P * a[:, None]).sum(axis=0)
        bb = b * A
        A = np.where(A == 0.0, 1.0, A)
        aa = ((P * a[:, None]) / A[None, :]).sum(axis=1)
        return aa, bb, t
    models = [cleanup(r) for r in res]
    a, b, t = models[bi]
    return dict(all_opt=objs, best_index=bi, alpha=a, beta=b, t=t, opt=objs[bi],
                all_alpha=np.array([m[0] for m in models]), all_beta=np.array([m[1] for m in models]),
                all_t=np.array([m[2] for m in models]))


def ref_alt(X, y, P, a0, b0, eta=0.0, eps=1e-6, T=100):
    Xo, Po = homogeneous(X, P)
    Xo, yo = regularize(Xo, y, Po, eta)
    a = np.array(a0, float); b = np.array(b0, float)
    loss = lambda a, b: np.linalg.norm(Xo @ ((Po * a[:, None]) @ b) - yo)
    old, opt, i = 1e20, 1e10, 1
    trace = []
    while i <= T and abs(old - opt) > eps * old:
        f = (Po * b[None, :]).sum(axis=1)
        a, _ = nnls(Xo * f[None, :], yo, maxiter=30 * Xo.shape[1])
        suma = (Po * a[:, None]).sum(axis=0)
        sumP = Po.sum(axis=0)
        for k in range(Po.shape[1]):
            if suma[k] == 0.0:
                a[Po[:, k] == 1] = 1.0 / sumP[k]
        suma = (Po * a[:, None]).sum(axis=0)
        a = a / (Po * suma[None, :]).sum(axis=1)
        b = b * suma
        Xa = Xo @ (Po * a[:, None])
        b = np.linalg.lstsq(Xa, yo, rcond=None)[0]
        old, opt = opt, loss(a, b)
        trace.append(opt)
        i += 1
    return dict(alpha=a[:-1], beta=b[:-1], t=b[-1] * a[-1], opt=opt, iters=i - 1, trace=np.array(trace))


def ref_bnb(X, y, P, eta=0.0):
    Xo, Po = homogeneous(X, P)
    Xo, yo = regularize(Xo, y, Po, eta)
    Mp = Xo.shape[1]

    def lower_bound(sig):
        Xp = Xo.copy(); Xm = -Xo.copy()
        Xp[:, sig < 0] = 0; Xm[:, sig > 0] = 0
        XX = np.hstack([Xp, Xm])
        aa, _ = nnls(XX, yo, maxiter=30 * XX.shape[1])
        ap = aa[:Mp].copy(); an = aa[Mp:].copy()
        ap[sig < 0] = 0; an[sig > 0] = 0
        return np.linalg.norm(XX @ aa - yo), ap - an

    def node(mu, sig):
        lb, a = lower_bound(sig)
        if lb >= mu:
            return np.inf, None, 1
        nu = np.zeros(Po.shape[1])
        for k in range(Po.shape[1]):
            idx = np.nonzero(Po[:, k])[0]
            for ii in range(len(idx)):
                for jj in range(ii + 1, len(idx)):
                    nu[k] += max(0.0, -a[idx[ii]] * a[idx[jj]])
        if np.all(nu == 0):
            return np.linalg.norm(Xo @ a - yo), a, 1
        k = int(np.argmax(nu))
        pk = Po[:, k] == 1
        sp = sig.copy(); sp[pk] = 1
        sm = sig.copy(); sm[pk] = -1
        mup, ap, np_ = node(mu, sp)
        mum, am, nm_ = node(min(mu, mup), sm)
        vals = [mu, mup, mum]; als = [a, ap, am]
        i = int(np.argmin(vals))
        return vals[i], als[i], np_ + nm_ + 1
    opt, a, nopen = node(np.inf, np.zeros(Mp, dtype=int))
    beta = (Po * a[:, None]).sum(axis=0)
    alpha = ((Po * a[:, None]) / beta[None, :]).sum(axis=1)
    return dict(alpha=alpha[:-1], beta=beta[:-1], t=beta[-1], opt=opt, nopen=nopen)


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, {k: np.shape(v) for k, v in kw.items()})


def main():
    # --- toy: the reference's own test input (test/runtests.jl:9-21) and known answers (runtests.jl:35-36) ---------
    X = np.array([[1., 2, 3], [3, 3, 4], [8, 1, 3], [5, 3, 1]])
    y = np.array([1., 1, 2, 3])
    P = np.array([[1, 0], [1, 0], [0, 1]], dtype=np.int64)
    r = ref_opt(X, y, P)
    assert r["best_index"] == 5 and r["opt"] < 1e-6
    ra = ref_alt(X, y, P, [.5, .5, .5, .5], [1, -1, 1])
    ra1 = ref_alt(X, y, P, [.5, .5, .5, .5], [1, 1, 1], T=1)
    rb = ref_bnb(X, y, P)
    save("toy", X=X, y=y, P=P,
         exact_alpha=np.array([5 / 11, 6 / 11, 1.0]), exact_beta=np.array([11 / 29, -16 / 29]), exact_t=60 / 29,
         opt_all_opt=r["all_opt"], opt_best_index=r["best_index"], opt_alpha=r["alpha"], opt_beta=r["beta"],
         opt_t=r["t"], opt_opt=r["opt"], opt_all_alpha=r["all_alpha"], opt_all_beta=r["all_beta"], opt_all_t=r["all_t"],
         alt_alpha0=np.array([.5, .5, .5, .5]), alt_beta0=np.array([1., -1, 1]),
         alt_alpha=ra["alpha"], alt_beta=ra["beta"], alt_t=ra["t"], alt_opt=ra["opt"],
         alt1_beta0=np.array([1., 1, 1]), alt1_opt=ra1["opt"], alt1_alpha=ra1["alpha"], alt1_beta=ra1["beta"], alt1_t=ra1["t"],
         bnb_alpha=rb["alpha"], bnb_beta=rb["beta"], bnb_t=rb["t"], bnb_opt=rb["opt"], bnb_nopen=rb["nopen"])

    # --- seeded synthetics (BASELINE.md §4 generator), sizes scipy finishes in seconds ----------------------------
    rng = np.random.default_rng(7)
    for name, (seed, N, D, K, eta) in {
        "synth_a": (20260101, 300, 12, 3, 0.0),
        "synth_b": (20260102, 500, 20, 5, 0.0),
        "synth_eta": (20260103, 400, 14, 4, 0.5),
        "synth_c": (20260104, 1200, 40, 6, 0.0),
    }.items():
        X, y, P, _ = O.synth(seed, N, D, K)
        X = np.ascontiguousarray(X)
        r = ref_opt(X, y, P, eta)
        a0 = rng.random(D + 1); b0 = (rng.random(K + 1) - 0.5) * 10
        ra = ref_alt(X, y, P, a0, b0, eta=eta)
        rb = ref_bnb(X, y, P, eta=eta)
        assert abs(rb["opt"] - r["opt"]) < 1e-8 * max(1, r["opt"]), (rb["opt"], r["opt"])
        save(name, seed=seed, eta=eta, X=X, y=y, P=P,
             opt_all_opt=r["all_opt"], opt_best_index=r["best_index"], opt_alpha=r["alpha"], opt_beta=r["beta"],
             opt_t=r["t"], opt_opt=r["opt"],
             alt_alpha0=a0, alt_beta0=b0, alt_alpha=ra["alpha"], alt_beta=ra["beta"], alt_t=ra["t"], alt_opt=ra["opt"],
             alt_iters=ra["iters"], alt_trace=ra["trace"],
             bnb_alpha=rb["alpha"], bnb_beta=rb["beta"], bnb_t=rb["t"], bnb_opt=rb["opt"], bnb_nopen=rb["nopen"])

    # --- a non-contiguous / unbalanced partition with correlated features (real-data-like conditioning) -----------
    rng = np.random.default_rng(11)
    N, D, K = 600, 16, 4
    Z = rng.standard_normal((N, 5))
    X = Z @ rng.standard_normal((5, D)) + 0.3 * rng.standard_normal((N, D))
    grp = rng.permutation(np.array([0] * 7 + [1] * 5 + [2] * 3 + [3] * 1))
    P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
    y = X @ (rng.random(D) * np.array([2., -3, 1, -1])[grp]) + 0.5 + 0.2 * rng.standard_normal(N)
    r = ref_opt(X, y, P)
    rb = ref_bnb(X, y, P)
    save("corr", X=X, y=y, P=P, opt_all_opt=r["all_opt"], opt_best_index=r["best_index"], opt_alpha=r["alpha"],
         opt_beta=r["beta"], opt_t=r["t"], opt_opt=r["opt"],
         bnb_alpha=rb["alpha"], bnb_beta=rb["beta"], bnb_t=rb["t"], bnb_opt=rb["opt"], bnb_nopen=rb["nopen"])


if __name__ == "__main__":
    main()
