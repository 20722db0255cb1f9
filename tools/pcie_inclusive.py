"""pcie_inclusive.py — the C3 fit through the HOST-pointer entry (partls_fit_opt: X uploaded inside the call), i.e. the rate a Julia
caller sees, next to the device-resident rate bench.py reports.  python tools/pcie_inclusive.py [reps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, partls_amd
from oracle import oracle as O
pls = partls_amd.package()
seed, N, D, K = 20260003, 100_000, 256, 20
X, y, P, _ = O.synth(seed, N, D, K)                       # bit-identical to the device generator
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pls.fit(pls.Opt, X, y, P)                                 # warm-up (allocations, first-touch)
t0 = time.perf_counter()
for _ in range(reps):
    m, _, rep = pls.fit(pls.Opt, X, y, P)
dt = (time.perf_counter() - t0) / reps
print(f"fit(Opt) from host memory (pageable numpy, {X.nbytes / 1e6:.0f} MB uploaded per call): {dt * 1e3:.2f} ms per fit = "
      f"{(1 << K) / dt / 1e6:.2f} M solves/s; opt {rep.opt:.12f} best_index {rep.best_index}")
mc = pls.MultiContext([0])
mc.fit_opt(X, y, P)
t0 = time.perf_counter()
for _ in range(reps):
    a, b, t, opt, bi, _ = mc.fit_opt(X, y, P)
dt = (time.perf_counter() - t0) / reps
print(f"partls_fit_opt_multi, 1 rank through RCCL: {dt * 1e3:.2f} ms per fit = {(1 << K) / dt / 1e6:.2f} M solves/s; opt {opt:.12f} best_index {bi}")
mc.close()
