"""BnB node rate of the register kernel against the deferred-update kernel around the switch-over: python tools/crossover_bnb.py N K cap D1 D2 ...
(target = intercept + noise, the search has to branch; flag PARTLS_OPT_GENERIC_KERNEL forces the n > 320 path)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
N, K, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pk = partls_amd.package(); ctx = pk.Context()
FAITHFUL, GEN = 1, 2                                    # PARTLS_OPT_FAITHFUL_INTERCEPT, PARTLS_OPT_GENERIC_KERNEL (include/partls.h)
dev = torch.device("cuda:0")
for D in (int(x) for x in sys.argv[4:]):
    P, wstar = pk.synth_truth(7, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device=dev); dy = torch.empty(N, dtype=torch.float64, device=dev)
    ctx.synth_device(7, N, D, np.zeros(D), dX.data_ptr(), dy.data_ptr())
    res = []
    for flags in (FAITHFUL, FAITHFUL | GEN):
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, flags)
        ctx.bnb_search(cap)
        t = time.perf_counter(); mu, pat, free, bounded = ctx.bnb_search(cap); dt = time.perf_counter() - t
        res.append((bounded / dt, mu, bounded))
    print(f"D={D} K={K}: register {res[0][0]:10.0f} nodes/s ({res[0][2]} nodes, incumbent {res[0][1]:.9f})   deferred {res[1][0]:10.0f} nodes/s ({res[1][2]} nodes, incumbent {res[1][1]:.9f})", flush=True)
