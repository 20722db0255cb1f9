"""Register kernel against the deferred-update kernel around the switch-over (n = 320): python tools/crossover_timing.py N K D1 D2 ...
Each D is swept twice per kernel (flag PARTLS_OPT_GENERIC_KERNEL forces the n > 320 path); prints solves/s of both."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
N, K = int(sys.argv[1]), int(sys.argv[2])
pk = partls_amd.package(); ctx = pk.Context()
GEN = 2                                                # PARTLS_OPT_GENERIC_KERNEL (include/partls.h)
dev = torch.device("cuda:0")
for D in (int(x) for x in sys.argv[3:]):
    P, wstar = pk.synth_truth(7, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device=dev); dy = torch.empty(N, dtype=torch.float64, device=dev)
    ctx.synth_device(7, N, D, wstar, dX.data_ptr(), dy.data_ptr())
    out = []
    for flags in (0, GEN):
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, flags)
        npat = ctx.num_patterns()
        for rep in range(2):
            obj, pat, _, unc = ctx.opt_sweep(0, npat)
        out.append((ctx.timing(2), obj, ctx.pivots(), unc))
    print(f"D={D} K={K} ({npat} patterns): register {npat / out[0][0] * 1e3:10.0f} solves/s ({out[0][0]:.2f} ms)   deferred {npat / out[1][0] * 1e3:10.0f} solves/s ({out[1][0]:.2f} ms)"
          f"   obj {out[0][1]:.9f} / {out[1][1]:.9f}  pivots {out[0][2]} / {out[1][2]}  unconv {out[0][3]} / {out[1][3]}", flush=True)
