#!/usr/bin/env python3
"""Sweep time of the full C3 range and of a 1/8 shard (what one rank of 8 runs) — run once per PARTLS_CHAIN_LEN setting."""
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
out = {"chain_len": os.environ.get("PARTLS_CHAIN_LEN", "default")}
for name, (lo, hi) in {"full": (0, npat), "shard_1_of_8": pls.dist.shard_range(npat, 3, 8), "shard_1_of_2": pls.dist.shard_range(npat, 1, 2)}.items():
    ts = []
    for _ in range(3):
        ctx.opt_sweep(lo, hi); ts.append(ctx.timing(L.T_SWEEP))
    out[name] = min(ts)
print(json.dumps(out))
ctx.close()
