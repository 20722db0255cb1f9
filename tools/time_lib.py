"""time_lib.py <lib.so> [patterns] — sweep time, pivots and cycles per pivot of one build of the library on the C3 problem (A/B and
timing-only experiment builds; select the build with the first argument)."""
import os, sys
os.environ['PARTLS_LIB'] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
pk = partls_amd.package(); ctx = pk.Context()
seed, N, D, K = 20260003, 100000, 256, 20
P, wstar = pk.synth_truth(seed, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device='cuda'); dy = torch.empty(N, dtype=torch.float64, device='cuda')
ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
for rep in range(2):
    r = ctx.opt_sweep(0, npat)
ms, piv = ctx.timing(2), ctx.pivots()
print(f"{sys.argv[1]}: {npat} patterns {ms:.3f} ms, {piv} pivots, unconverged {r[3]}, {ms * 1e-3 * 2.1e9 * 256 / max(piv, 1):.0f} CU-cycles per pivot (at 2.1 GHz)")
ctx.close()
