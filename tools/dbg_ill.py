import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import partls_amd
from oracle import oracle as O
pls = partls_amd.package(); O.build()
noise=1e-4
rng = np.random.default_rng(42)
N, D, K = 2000, 24, 4
Z = rng.standard_normal((N, 6))
X = Z @ rng.standard_normal((6, D)) + noise * rng.standard_normal((N, D))
grp = np.arange(D) % K
P = np.zeros((D, K), dtype=np.int64); P[np.arange(D), grp] = 1
y = X @ (rng.random(D) * np.array([1., -2, 3, -1])[grp]) + 0.3 + 0.05 * rng.standard_normal(N)
ref = O.fit_opt(X, y, P, return_all=True)
print("oracle opt", ref["opt"], ref["best_index"])
for flags, name in ((1, "blk faithful"), (3, "generic faithful")):
    ctx = pls.Context(); ctx.opt_prepare(X, y, P, 0.0, flags)
    obj, pat, allo, unc = ctx.opt_sweep(0, ctx.num_patterns(), want_all=True)
    err = np.abs(allo - ref["all_opt"]) / np.maximum(1, ref["all_opt"])
    print(name, "sweep obj", obj, "pat", pat, "unconv", unc, "max rel err", err.max(), "nbad", int((err > 1e-7).sum()), "pivots", ctx.pivots(), "vetoes", ctx.vetoes())
    bad = np.nonzero(err > 1e-7)[0][:8]
    for b in bad: print("   pattern", b, "gpu", allo[b], "oracle", ref["all_opt"][b])
for flags, name in ((0, "blk free"), (2, "generic free")):
    ctx = pls.Context(); ctx.opt_prepare(X, y, P, 0.0, flags)
    obj, pat, allo, unc = ctx.opt_sweep(0, ctx.num_patterns())
    a, b, t, o, bi = ctx.opt_finish(pat)
    print(name, "sweep obj", obj, "pat", pat, "unconv", unc, "finish opt", o, "bi", bi, "pivots", ctx.pivots(), "vetoes", ctx.vetoes())
