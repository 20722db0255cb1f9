#!/usr/bin/env python3
"""The calibration's view of one synthetic problem: measured pivots per flip of every group, the order it picks, the pivots per
pattern the additive model predicts for the reference's order and for the picked one, and the calibration time.
usage: bit_order_costs.py N D K [seed]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, partls_amd
pls = partls_amd.package(); L = pls.lowlevel
N, D, K = (int(x) for x in sys.argv[1:4]); seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
ctx = pls.Context(0)
P, ws = pls.synth_truth(seed, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
gbit, cost = ctx.bit_order()
w = 0.5 ** (np.arange(len(cost)) + 1)
print(json.dumps(dict(N=N, D=D, K=K, seed=seed, calib_ms=ctx.timing(L.T_CALIB), gbit=gbit.tolist(), cost=[round(float(c), 2) for c in cost],
                      predicted_identity=float(w @ cost), predicted_sorted=float(w @ np.sort(cost)))))
ctx.close()
