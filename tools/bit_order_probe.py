#!/usr/bin/env python3
"""Does the assignment of groups to Gray-code bits matter?  Bit b of the Gray index flips in 2^-(b+1) of all transitions, so the
group on the fastest bit pays half of all exchanges.  Same C3 problem, columns of P permuted (the set of subproblems is the same,
only the visiting order changes): sweep time and pivots per order.  Usage: bit_order_probe.py [config]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, partls_amd
pls = partls_amd.package(); L = pls.lowlevel
cfg = {"C3": (20260003, 100_000, 256, 20), "C2": (20260002, 10_000, 128, 12)}[sys.argv[1] if len(sys.argv) > 1 else "C3"]
seed, N, D, K = cfg
ctx = pls.Context(0)
P, ws = pls.synth_truth(seed, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()


def run(perm):
    Pp = np.asfortranarray(P[:, perm])
    ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, Pp, 0.0, 0)
    npat = ctx.num_patterns()
    ts = []
    for _ in range(2):
        bobj, bpat, _, unconv = ctx.opt_sweep(0, npat); ts.append(ctx.timing(L.T_SWEEP))
    return dict(ms=min(ts), pivots=int(ctx.pivots()), obj=float(bobj), unconv=int(unconv))


if len(sys.argv) > 2 and sys.argv[2] == "calib":       # what the library does by itself (PARTLS_BIT_ORDER unset / calibrate / identity)
    for rep in range(2):
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, np.asfortranarray(P), 0.0, 0)
        gbit, cost = ctx.bit_order()
        npat = ctx.num_patterns()
        bobj, bpat, _, unconv = ctx.opt_sweep(0, npat)
        print(json.dumps(dict(order="library", gbit=gbit.tolist(), flip_cost=[round(float(x), 2) for x in cost], calib_ms=ctx.timing(L.T_CALIB),
                              ms=ctx.timing(L.T_SWEEP), pivots=int(ctx.pivots()), obj=float(bobj), pat=int(bpat), unconv=int(unconv))), flush=True)
    ctx.close()
    sys.exit(0)
ident = np.arange(K)
base = run(ident)
print(json.dumps(dict(order="identity", **base)), flush=True)
print(json.dumps(dict(order="reversed", **run(ident[::-1]))), flush=True)
# group k on the first column / on the last column, the rest in place: which end is the fast bit, and what does group k cost there?
for k in range(K):
    first = np.concatenate([[k], np.delete(ident, k)])
    last = np.concatenate([np.delete(ident, k), [k]])
    print(json.dumps(dict(order=f"group {k} first", **run(first))), flush=True)
    print(json.dumps(dict(order=f"group {k} last", **run(last))), flush=True)
rng = np.random.default_rng(5)
for i in range(4):
    print(json.dumps(dict(order=f"random {i}", **run(rng.permutation(K)))), flush=True)
ctx.close()
