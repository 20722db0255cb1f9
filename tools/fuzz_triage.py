"""Arbitration of a fuzz mismatch by KKT certificates (necessary and sufficient for the convex sign-constrained problem):
for one problem of tests/test_gpu_fuzz.py, every pattern on which the HIP path and the C oracle disagree is examined: the
oracle's solution, and the HIP path's solution (partls_opt_pattern), each with x >= 0, gradient on the zero set <= 0, gradient on
the support = 0, and the objective recomputed from the data."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import partls_amd
from oracle import oracle as O
from test_gpu_fuzz import _random_problem
block, it = int(sys.argv[1]), int(sys.argv[2]); base = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
pls = partls_amd.package(); O.build()
rng = np.random.default_rng(base + block)
for i in range(it + 1):
    X, y, P, eta = _random_problem(rng)
N, M = X.shape; K = P.shape[1]
print("shape", X.shape, "K", K, "eta", eta, "rank", np.linalg.matrix_rank(X), "cond %.3g" % np.linalg.cond(X),
      "col norm range %.3g %.3g" % (np.linalg.norm(X, axis=0).min(), np.linalg.norm(X, axis=0).max()), "group sizes", P.sum(0))
ref = O.fit_opt(X, y, P, eta=eta, return_all=True)
ctx = pls.Context(); ctx.opt_prepare(X, y, P, eta, 1)
obj, pat, got, unc = ctx.opt_sweep(0, ctx.num_patterns(), want_all=True)
Xo = np.column_stack([X, np.ones(N)]); Po = np.zeros((M + 1, K + 1), dtype=np.int64); Po[:M, :K] = P; Po[M, K] = 1
yo = y
if eta != 0.0:
    Xo = np.vstack([Xo, np.sqrt(eta) * Po.T.astype(float)]); yo = np.concatenate([y, np.zeros(K + 1)])
bad = np.nonzero(np.abs(got - ref["all_opt"]) > 1e-8 * np.maximum(1, ref["all_opt"]) + 2e-7 * max(1, np.linalg.norm(y)))[0]
print("mismatched patterns:", bad.tolist(), "unconverged", unc)
def kkt(A, x):
    r = yo - A @ x; g = A.T @ r; cn = np.linalg.norm(A, axis=0) + 1e-300
    z = x == 0
    return np.linalg.norm(r), x.min(), (g[z] / cn[z]).max() if z.any() else 0.0, (np.abs(g[~z]) / cn[~z]).max() if (~z).any() else 0.0
for b in bad[:6]:
    s = np.array([1.0 if (b >> k) & 1 else -1.0 for k in range(K + 1)])
    A = Xo * (Po @ s)[None, :]
    xo, rno, mode, ns = O.nnls(A, yo)
    ra, og = ctx.opt_pattern(int(b))
    print(f"pattern {b}: sweep {got[b]:.10g} | gpu re-solve {og:.10g} kkt(resid, min x, max g zeros, max|g| supp) = {tuple(float('%.3g' % v) for v in kkt(A, ra))}"
          f" | oracle {ref['all_opt'][b]:.10g} kkt = {tuple(float('%.3g' % v) for v in kkt(A, xo))}")
for cl in ("1", "2", "4", "16"):
    os.environ["PARTLS_CHAIN_LEN"] = cl
    c2 = pls.Context(); c2.opt_prepare(X, y, P, eta, 1)
    o2, p2, g2, u2 = c2.opt_sweep(0, c2.num_patterns(), want_all=True)
    nb = int((np.abs(g2 - ref["all_opt"]) > 1e-8 * np.maximum(1, ref["all_opt"]) + 2e-7 * max(1, np.linalg.norm(y))).sum())
    print("chain_len", cl, "mismatches", nb, "pivots", c2.pivots())
os.environ.pop("PARTLS_CHAIN_LEN")
for tr in ("1e-11", "1e-9"):
    pass
