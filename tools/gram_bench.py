#!/usr/bin/env python3
"""Gram-build timing on a BASELINE-shaped problem (default C4: N=1M, D=512): HIP-event ms and useful/executed TFLOP/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, partls_amd
pls = partls_amd.package(); L = pls.lowlevel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = pls.Context(0)
P, ws = pls.synth_truth(20260004, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
ctx.synth_device(20260004, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
ms = []
for _ in range(reps):
    ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
    ms.append(ctx.timing(L.T_GRAM))
na = D + 2
useful = 2.0 * N * na * (na + 1) / 2
print(json.dumps({"N": N, "D": D, "gram_ms": ms, "best_ms": min(ms), "useful_tflops": useful / (min(ms) * 1e-3) / 1e12}))
if os.environ.get("GRAM_CHECK"):
    G = ctx.gram()
    Xh = dX.cpu().numpy().reshape(D, N).T; yh = dy.cpu().numpy()
    Z = np.hstack([Xh, np.ones((N, 1)), yh[:, None]])
    ref = Z.T @ Z
    print("max rel err", np.abs(G - ref).max() / np.abs(ref).max())
