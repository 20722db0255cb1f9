"""host_overhead.py [config] — wall-clock time of each staged call of one fit(Opt) step against the HIP-event times of the kernels in it
(what the host adds around the device work: argument checks, mask / permutation building, copies, stream syncs, ctypes)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, partls_amd
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
kind, seed, N, D, K = bench.CONFIGS[cfg]
pls = partls_amd.package(); L = pls.lowlevel; ctx = pls.Context(0)
P, wstar = pls.synth_truth(seed, D, K)
dev = torch.device("cuda:0")
dX = torch.empty(N * D, dtype=torch.float64, device=dev); dy = torch.empty(N, dtype=torch.float64, device=dev)
ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr()); torch.cuda.synchronize()
acc = {"prepare": 0.0, "sweep": 0.0, "finish": 0.0}; gpu = {"gram": 0.0, "prep": 0.0, "calib": 0.0, "sweep": 0.0, "finish": 0.0}
reps = 200 if cfg == "C2" else 10
for it in range(reps + 5):
    t0 = time.perf_counter(); ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
    t1 = time.perf_counter(); bo, bp, _, unc = ctx.opt_sweep(0, -1)
    t2 = time.perf_counter(); a, b, t, opt, bi = ctx.opt_finish(bp)
    t3 = time.perf_counter()
    if it >= 5:
        acc["prepare"] += t1 - t0; acc["sweep"] += t2 - t1; acc["finish"] += t3 - t2
        for k, w in (("gram", L.T_GRAM), ("prep", L.T_PREP), ("calib", L.T_CALIB), ("sweep", L.T_SWEEP), ("finish", L.T_FINISH)): gpu[k] += ctx.timing(w)
print(cfg, "wall us per call:", {k: round(v / reps * 1e6, 1) for k, v in acc.items()}, "total", round(sum(acc.values()) / reps * 1e6, 1))
print(cfg, "HIP-event us:", {k: round(v / reps * 1e3, 1) for k, v in gpu.items()})
