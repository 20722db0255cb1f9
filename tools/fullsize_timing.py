#!/usr/bin/env python3
"""Wall-clock of the secondary BASELINE configs at full size on one MI355X (C4: Alt N=1M D=512 K=16, T=200; C5: BnB N=100k
D=256 K=24, plus the 2^24-pattern Opt sweep it is checked against).  Not a bench line: for DESIGN.md §6."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import partls_amd
pls = partls_amd.package(); L = pls.lowlevel
ctx = pls.Context(0)          # module global ON PURPOSE: round 1 crashed in exit() with this + the profiler (api.py atexit hook)

def problem(seed, N, D, K):
    P, ws = pls.synth_truth(seed, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device="cuda"); dy = torch.empty(N, dtype=torch.float64, device="cuda")
    ctx.synth_device(seed, N, D, ws, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
    return dX, dy, P

out = {}
# ---- C5: BnB vs Opt at K=24
seed, N, D, K = 20260005, 100_000, 256, 24
dX, dy, P = problem(seed, N, D, K)
t0 = time.perf_counter(); ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0); bo, bp, _, unc = ctx.opt_sweep(0, -1)
a, b, t, oo, bi = ctx.opt_finish(bp); t1 = time.perf_counter()
out["C5_opt_2^24"] = dict(seconds=t1 - t0, sweep_ms=ctx.timing(L.T_SWEEP), opt=oo, solves_per_s=(1 << K) / (ctx.timing(L.T_SWEEP) * 1e-3))
t0 = time.perf_counter(); ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
ab, bb, tb, ob, nopen = ctx.bnb_prepared(); t1 = time.perf_counter()
out["C5_bnb"] = dict(seconds=t1 - t0, nopen=nopen, opt=ob, gap_vs_opt=abs(ob - oo) / oo)
del dX, dy; torch.cuda.empty_cache()
# ---- C4: Alt, N=1M, D=512, K=16, T=200
seed, N, D, K = 20260004, 1_000_000, 512, 16
dX, dy, P = problem(seed, N, D, K)
rng = np.random.default_rng(123); a0 = rng.random(D + 1); b0 = (rng.random(K + 1) - 0.5) * 10
t0 = time.perf_counter(); ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT); t1 = time.perf_counter()
gram_ms = ctx.timing(L.T_GRAM)
a, b, t, opt, iters = ctx.alt_prepared(a0, b0, eps=1e-6, T=200); t2 = time.perf_counter()
flops = 2.0 * N * (D + 2) ** 2 / 2
# the same start with eps = 1e-300: iterates until the objective stops changing exactly or T = 200 (steady-state cost per ALS iteration)
t3 = time.perf_counter(); a2, b2, tt2, opt2, iters2 = ctx.alt_prepared(a0, b0, eps=1e-300, T=200); t4 = time.perf_counter()
out["C4_alt_200_iterations"] = dict(alt_s=t4 - t3, iters=iters2, ms_per_iteration=(t4 - t3) * 1e3 / max(1, iters2), opt=opt2)
out["C4_alt"] = dict(prepare_s=t1 - t0, gram_ms=gram_ms, gram_tflops_useful=flops / (gram_ms * 1e-3) / 1e12, alt_s=t2 - t1, iters=iters, opt=opt,
                     noise_floor=0.1 * np.sqrt(N))
print(json.dumps(out, indent=1))
# teardown experiments for the exit-time crash under rocprofv3 (profiles/README.md): FS_MODE=close closes the context explicitly,
# FS_MODE=free additionally drops the torch tensors and the caching allocator's blocks before the interpreter exits
mode = os.environ.get("FS_MODE", "")
if mode in ("close", "free"):
    ctx.close()
if mode == "free":
    del dX, dy
    torch.cuda.empty_cache()
