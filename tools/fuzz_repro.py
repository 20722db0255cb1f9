"""Reproduce one problem of tests/test_gpu_fuzz.py (block, it) and print where the HIP path and the oracle part ways."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import partls_amd
from oracle import oracle as O
from test_gpu_fuzz import _random_problem
block, it, base = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 9500
pls = partls_amd.package(); O.build()
rng = np.random.default_rng(base + block)
for i in range(it + 1):
    X, y, P, eta = _random_problem(rng)
print("shape", X.shape, "K", P.shape[1], "eta", eta, "col norms min/max", np.linalg.norm(X, axis=0).min(), np.linalg.norm(X, axis=0).max(),
      "rank", np.linalg.matrix_rank(X), "cond", np.linalg.cond(X))
ref = O.fit_opt(X, y, P, eta=eta, return_all=True)
print("oracle opt", ref["opt"], "best", ref["best_index"])
for faithful in (False, True):
    ctx = pls.Context()
    ctx.opt_prepare(X, y, P, eta, 1 if faithful else 0)
    obj, pat, allo, unc = ctx.opt_sweep(0, ctx.num_patterns(), want_all=faithful)
    a, b, t, o, bi = ctx.opt_finish(pat)
    print("faithful", faithful, "sweep obj", obj, "pattern", pat, "unconv", unc, "| finish opt", o, "best_index", bi, "t", t)
    if faithful:
        d = np.abs(allo - ref["all_opt"]); print("   max |all_opt diff|", d.max(), "at", int(d.argmax()))
K = P.shape[1]
# oracle objective restricted to intercept-sign-free patterns: min over the two intercept signs
ao = ref["all_opt"]; lo = np.minimum(ao[: 1 << K], ao[1 << K:])
print("oracle min over intercept signs per pattern: best", lo.min(), "at", int(lo.argmin()))
ctx = pls.Context(); ctx.opt_prepare(X, y, P, eta, 1)
G = ctx.gram()
Z = np.column_stack([X, np.ones(len(y)), y]); Gn = Z.T @ Z
rel = np.abs(G - Gn) / (np.sqrt(np.outer(np.diag(Gn), np.diag(Gn))) + 1e-300)
print("gram max scaled diff", rel.max(), "at", np.unravel_index(rel.argmax(), rel.shape), "nan in G:", np.isnan(G).any())
print("group sizes", P.sum(0), "contiguous groups:", bool(np.all(np.diff(P.argmax(1)) >= 0)))
# which variables carry the worst per-pattern error?
sizes = P.sum(0)
for name, Xv in (("unit column scale", X / np.linalg.norm(X, axis=0)), ("sorted groups", X[:, np.argsort(P.argmax(1), kind="stable")])):
    Pv = P if name == "unit column scale" else P[np.argsort(P.argmax(1), kind="stable")]
    r2 = O.fit_opt(Xv, y, Pv, eta=eta, return_all=True)
    c2 = pls.Context(); c2.opt_prepare(Xv, y, Pv, eta, 1)
    o2, p2, a2, u2 = c2.opt_sweep(0, c2.num_patterns(), want_all=True)
    print(name, ": max |all_opt diff|", np.abs(a2 - r2["all_opt"]).max(), "gpu best", o2, "oracle", r2["opt"])
