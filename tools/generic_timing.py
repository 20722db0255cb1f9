"""Sweep timing at any D (register kernel up to n = 320, sweep_generic.hip beyond): python tools/generic_timing.py N D K"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
N, D, K = (int(x) for x in sys.argv[1:4])
pk = partls_amd.package(); ctx = pk.Context()
P, wstar = pk.synth_truth(7, D, K)
dev = torch.device("cuda:0")
dX = torch.empty(N * D, dtype=torch.float64, device=dev); dy = torch.empty(N, dtype=torch.float64, device=dev)
ctx.synth_device(7, N, D, wstar, dX.data_ptr(), dy.data_ptr())
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
npat = ctx.num_patterns()
for rep in range(2):
    obj, pat, _, unc = ctx.opt_sweep(0, npat)
print(f"N={N} D={D} K={K}: {npat} patterns, sweep {ctx.timing(2):.2f} ms, {npat / ctx.timing(2) * 1e3:.0f} solves/s, pivots {ctx.pivots()}, "
      f"{ctx.timing(2) * 1e3 / max(1, ctx.pivots()):.2f} us/pivot (all workgroups), obj {obj:.6f} unconv {unc}")
