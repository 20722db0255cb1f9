"""numpy emulation of the device sweep's DECISIONS (development aid; never imported by the product or the tests).

Follows sweep_blk.hip / sweep_generic.hip step by step — unit-diagonal scaled Gram tableau, Gray-code chains, KKT scan, block
principal pivots by 16-wide tile column in blocks of <= 8, the dependent-column rejection and the backup rule — with sequential
rank-1 sweeps on a dense symmetric matrix (the blocked panel evaluation of the kernels differs in round-off only).  Used to
study rank-deficient fuzz cases on the CPU before spending GPU minutes:  python tools/tableau_emul.py <block> <it> [base]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))


def prepare(X, y, P, eta, faithful=True):
    N, M = X.shape
    K = P.shape[1]
    Z = np.column_stack([X, np.ones(N), y])
    G = Z.T @ Z
    mask = np.zeros(M + 2, dtype=np.uint64)
    for k in range(K):
        mask[:M] |= (P[:, k].astype(np.uint64) << np.uint64(k))
    mask[M] = np.uint64(1) << np.uint64(K)
    if eta:
        for a in range(M + 1):
            for b in range(M + 1):
                G[a, b] += eta * bin(int(mask[a] & mask[b])).count("1")
    n = M + 1 if faithful else M
    key = [(int(mask[v]) & -int(mask[v])).bit_length() - 1 if mask[v] else 64 for v in range(n)]
    perm = sorted(range(n), key=lambda v: key[v])
    idx = perm + [M + 1]
    B = G[np.ix_(idx, idx)].copy()
    if not faithful:
        gI = G[M, idx]
        B -= np.outer(gI, gI) / G[M, M]
    d = np.diag(B)[:n]
    ref = np.array([G[p, p] for p in perm])
    scale = np.where((d > 0) & (d > 1e-14 * np.abs(ref)), 1.0 / np.sqrt(np.where(d > 0, d, 1.0)), 0.0)
    s = np.concatenate([scale, [1.0]])
    T = B * s[:, None] * s[None, :]
    for i in range(n):
        if scale[i] == 0.0:
            T[i, i] = 1.0
    return dict(T0=T, n=n, mask=mask[perm], kbits=K + 1 if faithful else K, perm=perm, scale=scale,
                tol=1e-11 * np.sqrt(max(G[M + 1, M + 1], 0.0)), G=G)


def pivot(T, k, d):
    inv = 1.0 / d
    col = T[:, k].copy()
    T -= np.outer(col, col) * inv
    T[:, k] = col * abs(inv)
    T[k, :] = col * abs(inv)
    T[k, k] = -inv


class Policy:
    """entering-pivot acceptance rule.  fixed: d > eps.  refined: recompute d from the pristine Gram when the tableau value is
    inside the round-off band of the chain (growth-scaled)."""

    def __init__(self, kind="fixed", eps=1e-11, band=1e-13):
        self.kind, self.eps, self.band = kind, eps, band
        self.nrefined = 0

    def accept(self, st, k):
        T, T0, basic, n = st["T"], st["T0"], st["basic"], st["n"]
        d = T[k, k]
        if self.kind == "fixed":
            return d > self.eps
        if d <= self.eps:
            return False
        if self.kind == "loo":
            # leave-one-out rule: entering k must not leave ANY basic column dependent on the others at the eps level:
            # 1/d_j(S) = 1/d_j(B) + c_j^2/d_k  with c_j = T[j,k]
            col = np.delete(T[:n, k], k)          # every variable row: nonbasic rows satisfy T_jk^2 <= d_k (PSD Schur complement)
            cmax2 = float(np.max(col ** 2)) if len(col) else 0.0
            if d <= self.eps * max(1.0, cmax2):
                self.nrefined += 1
                st["log"].append((k, d, cmax2, st["growth"]))
                return False
            return True
        if d > self.band * st["growth"]:
            return True
        # refined pivot from the pristine Gram: d = G_kk - 2 c'G_Bk + c' G_BB c with c = T[B, k] (regression coefficients)
        self.nrefined += 1
        Bset = np.nonzero(basic)[0]
        c = T[Bset, k]
        r = T0[np.ix_(list(Bset) + [k], [k])][:, 0] - T0[np.ix_(list(Bset) + [k], Bset)] @ c
        dref = r[-1] - c @ r[:-1]
        st["log"].append((k, d, dref, st["growth"]))
        return dref > self.eps


def sweep(pr, policy, chain_len=None, verbose=False):
    n, T0, mask, kbits = pr["n"], pr["T0"], pr["mask"], pr["kbits"]
    npat = 1 << kbits
    if chain_len is None:
        chain_len = 512
        while chain_len > 16 and (npat + chain_len - 1) // chain_len < 512:
            chain_len >>= 1
    out = np.zeros(npat)
    tol = pr["tol"]
    pop = lambda v: bin(int(v)).count("1")
    log = []
    for g0 in range(0, npat, chain_len):
        T = T0.copy()
        basic = np.zeros(n, dtype=bool)
        st = dict(T=T, T0=T0, basic=basic, n=n, growth=1.0, log=log)
        for g in range(g0, min(g0 + chain_len, npat)):
            pat = g ^ (g >> 1)
            f = np.array([2 * pop(int(mask[v]) & pat) - pop(mask[v]) for v in range(n)])
            blocked = np.zeros(n, dtype=bool)
            ninf_best, patience, rounds, progress = n + 1, 3, 0, False
            while True:
                if progress:
                    blocked[:] = False
                progress = False
                q = T[:n, n]
                fq = np.where(f > 0, q, np.where(f < 0, -q, 0.0))
                bad = np.where(basic, (f == 0) | (fq < -tol), (fq > tol) & ~blocked)
                viol = np.nonzero(bad)[0]
                if len(viol) == 0:
                    break
                if len(viol) < ninf_best:
                    ninf_best, patience, allv = len(viol), 3, True
                elif patience > 0:
                    patience -= 1
                    allv = True
                else:
                    allv = False
                rounds += 1
                if rounds > 20 * (n + 1):
                    break
                if not allv:
                    viol = viol[-1:]
                for k in viol:                    # ascending; tile columns / blocks of 8 are sequential in this order
                    if basic[k]:
                        d = T[k, k]
                        pivot(T, k, d)
                        st["growth"] = max(st["growth"], abs(1.0 / d))
                        basic[k] = False
                        progress = True
                    elif policy.accept(st, k):
                        d = T[k, k]
                        pivot(T, k, d)
                        st["growth"] = max(st["growth"], abs(1.0 / d))
                        basic[k] = True
                        progress = True
                    else:
                        blocked[k] = True
            out[pat] = np.sqrt(max(T[n, n], 0.0))
    return out, log


if __name__ == "__main__":
    from oracle import oracle as O
    from test_gpu_fuzz import _random_problem
    block, it = int(sys.argv[1]), int(sys.argv[2])
    base = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    O.build()
    rng = np.random.default_rng(base + block)
    for i in range(it + 1):
        X, y, P, eta = _random_problem(rng)
    print("shape", X.shape, "K", P.shape[1], "eta", eta, "rank", np.linalg.matrix_rank(X), "cond %.3g" % np.linalg.cond(X))
    ref = O.fit_opt(X, y, P, eta=eta, return_all=True)["all_opt"]
    pr = prepare(X, y, P, eta)
    for pol in (Policy("fixed"), Policy("loo")):
        got, log = sweep(pr, pol)
        err = np.abs(got - ref) / np.maximum(1.0, ref)
        print(pol.kind, "max rel err %.3g" % err.max(), "mismatches", int((err > 1e-8).sum()), "refined tests", pol.nrefined)
        for e in log[:12]:
            print("   var %d tableau d %.3g refined d %.3g growth %.3g" % e)
