#!/usr/bin/env python3
"""Does the variable -> tile layout matter once the groups sit on the Gray bits by flip cost?  The layout follows the reference's
group order (ctx_prepare: stable sort on the lowest group), so permuting the columns of P changes the layout and nothing else —
the library's calibration finds the same fast groups either way.  C3, run on the GPU box."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
pls = partls_amd.package(); L = pls.lowlevel
seed, N, D, K = 20260003, 100_000, 256, 20
ctx = pls.Context(0)
P, ws = pls.synth_truth(seed, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
fast = [19,17,18,9,10,2,5,12,13,7,4,15,6,3,1,11,16,0,14,8]
rng = np.random.default_rng(3)
orders = {"identity": list(range(K)), "fast groups first in the layout": fast, "fast groups last in the layout": fast[::-1],
          "random a": rng.permutation(K).tolist(), "random b": rng.permutation(K).tolist()}
for name, perm in orders.items():
    Pp = np.asfortranarray(P[:, perm])
    ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, Pp, 0.0, 0)
    gbit, cost = ctx.bit_order()
    ts = []
    for _ in range(2):
        bobj, bpat, _, unconv = ctx.opt_sweep(0, -1); ts.append(ctx.timing(L.T_SWEEP))
    print(json.dumps(dict(layout=name, ms=min(ts), pivots=int(ctx.pivots()), fastest=[perm[int(k)] for k in np.argsort(gbit)[:4]])), flush=True)
ctx.close()
