import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, partls_amd
pls = partls_amd.package(); L = pls.lowlevel
seed, N, D, K = 20260003, 100_000, 256, 20
ctx = pls.Context(0)
P, ws = pls.synth_truth(seed, D, K)
dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
npat = ctx.num_patterns()
out = {}
for name, (lo, hi) in {"full": (0, npat), "third": pls.dist.shard_range(npat, 1, 3), "fifth": pls.dist.shard_range(npat, 2, 5), "eighth": pls.dist.shard_range(npat, 3, 8), "odd": (12345, 12345 + 700001)}.items():
    ts = []
    for _ in range(2):
        r = ctx.opt_sweep(lo, hi); ts.append(ctx.timing(L.T_SWEEP))
    out[name] = (round(min(ts), 3), round((hi - lo) / min(ts) / 1e3, 2))
print(json.dumps(out))
parts = [ctx.opt_sweep(*pls.dist.shard_range(npat, r, 3)) for r in range(3)]
full = ctx.opt_sweep(0, npat)
best = min((p[0], p[1]) for p in parts)
print("3-shard winner", best[1] == full[1], abs(best[0] - full[0]) <= 1e-10 * full[0])
ctx.close()
