#!/usr/bin/env python3
"""bench.py — sign-pattern NNLS solves/sec of fit(Opt) on the BASELINE.json headline config, on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over the synthetic problem, inputs resident in HBM when the timed region starts:
Gram build (fp64 MFMA) -> tableau prep -> sign-pattern sweep over this rank's shard of the Gray-index space ->
all-reduce(min objective, then min index among the minimisers) over RCCL -> winner re-solve + objective from the data.
The 2^K' patterns of ONE problem are sharded across ranks (total work fixed => "strong" scaling); there is no data-path
collective besides the two 8-byte all-reduces.  value = patterns solved by all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (see README/DESIGN.md §6 for the roofline and cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (seed, N, D, K)   — BASELINE.md §4
    "C2": (20260002, 10_000, 128, 12),
    "C3": (20260003, 100_000, 256, 20),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # public datasheet (vector = matrix fp64 on MI355X); not in the on-image guides


def algorithmic_bytes_per_solve(Mp):
    """read G + c_f + scratch x once, write the objective: 8*(M'^2 + 2M') + 8 bytes."""
    return 8 * (Mp * Mp + 2 * Mp) + 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--faithful", action="store_true", help="enumerate the reference's 2^(K+1) patterns (intercept sign too)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-patterns", type=int, default=0, help="patterns in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal hooks (never used by the driver): PARTLS_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # PARTLS_DIST_BACKEND=gloo swaps the transport, so the N>1 control flow can be exercised on a one-GPU box
    share = os.environ.get("PARTLS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("PARTLS_DIST_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    import partls_amd
    pls = partls_amd.package()
    L = pls.lowlevel
    ctx = pls.Context(dev_index)

    seed, N, D, K = CONFIGS[args.config]
    P, wstar = pls.synth_truth(seed, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device=dev)         # column-major N x D, resident in HBM
    dy = torch.empty(N, dtype=torch.float64, device=dev)
    ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    flags = L.OPT_FAITHFUL_INTERCEPT if args.faithful else 0

    def step():
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, flags)
        npat = ctx.num_patterns()
        g0, g1 = pls.dist.shard_range(npat, rank, world)
        bobj, bpat, _, unconv = ctx.opt_sweep(g0, g1)
        t_gram, t_prep, t_sweep = ctx.timing(L.T_GRAM), ctx.timing(L.T_PREP), ctx.timing(L.T_SWEEP)
        pivots = ctx.pivots()
        # all-reduce(min residual), then min pattern index among the minimisers (first-index argmin, Opt.jl:96)
        _, bpat = pls.dist.allreduce_argmin(bobj, bpat, device=red_dev)
        a, b, t, opt, bi = ctx.opt_finish(bpat)
        return dict(npat=npat, local=g1 - g0, opt=opt, best_index=bi, unconv=unconv, t_gram=t_gram, t_prep=t_prep,
                    t_sweep=t_sweep, t_finish=ctx.timing(L.T_FINISH), alpha=a, beta=b, t=t, pivots=pivots)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    sweep_ms, gram_ms = [], []
    res = None
    for _ in range(args.steps):
        res = step()
        sweep_ms.append(res["t_sweep"])
        gram_ms.append(res["t_gram"])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    npat = res["npat"]
    value = npat * args.steps / elapsed
    Mp = D + 1
    # dominant kernel = the sweep; its launch processes this rank's shard; duration from HIP events on the library's stream
    sweep_avg_s = (sum(sweep_ms) / len(sweep_ms)) * 1e-3
    solves_per_launch = res["local"]
    bytes_per_solve = algorithmic_bytes_per_solve(Mp)
    achieved_gbs = solves_per_launch * bytes_per_solve / sweep_avg_s / 1e9
    # fp64 view of the same kernel: every pivot is one rank-1 update of the symmetric tableau = T(T+1)/2 tile slots x 256 FMAs
    n_tab = Mp if args.faithful else D
    tiles = (n_tab + 15) // 16
    flops = res["pivots"] * (tiles * (tiles + 1) // 2) * 256 * 2
    fp64_tflops = flops / sweep_avg_s / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_sweep_traffic.json")
    if world == 1 and args.config == "C3" and not args.faithful and os.path.exists(tpath):
        traffic = json.load(open(tpath))["hbm_bytes_per_launch"]       # rocprofv3 PMC, measured offline on this kernel
    out = {
        "metric": "sign-pattern NNLS solves/sec (whole node); fp64 obj gap vs ref",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: fit(Opt) N={N} D={D} K={K}, {npat} sign patterns "
                               f"({'2^(K+1) faithful' if args.faithful else '2^K, free intercept'}), eta=0",
                   "seed": seed, "sharding": f"gray-index range / {world} ranks"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "sweep_blk_kernel", "kernel_ms": sweep_avg_s * 1e3,
                     "algorithmic_bytes_per_solve": bytes_per_solve, "solves_per_launch": solves_per_launch},
        "roofline_fp64": {"bound": "fp64 vector FMA", "achieved": fp64_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": fp64_tflops / FP64_PEAK_TFLOPS, "pivots_per_launch": res["pivots"],
                          "note": "the tableau is register-resident, so the kernel's real bound is fp64 FMA issue, not HBM"},
        "kernels_ms": {"gram_build": sum(gram_ms) / len(gram_ms), "prep": res["t_prep"], "sweep": sweep_avg_s * 1e3,
                       "finish": res["t_finish"]},
        "result": {"opt": res["opt"], "best_index": res["best_index"], "unconverged": res["unconv"]},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, seed, N, D, K, res, np)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, seed, N, D, K, res, np):
    """The oracle in reference-faithful dense mode (Opt.jl:87-90 per pattern: column scaling + dense Lawson–Hanson on the
    N x (D+1) matrix + objective), single-threaded like the reference, on a bounded sample of patterns that always
    includes the GPU's winner (whose objective is compared: the fp64 objective gap)."""
    from oracle import oracle as O
    X, y, P, _ = O.synth(seed, N, D, K)
    Xo, Po = O.homogeneous(X, P)
    npat_ref = 1 << (K + 1)
    n = args.cpu_patterns or (3 if N * D > 5_000_000 else 24)
    rng = np.random.default_rng(0)
    sample = [int(res["best_index"])] + [int(v) for v in rng.integers(0, npat_ref, size=n - 1)]
    t0 = time.perf_counter()
    objs = O.opt_patterns(Xo, y, Po, np.array(sample, dtype=np.int64))
    dt = time.perf_counter() - t0
    gap = abs(objs[0] - res["opt"]) / max(1.0, objs[0])
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{n} of {npat_ref} patterns (winner + random), dense Lawson-Hanson per pattern as Opt.jl:87-90, "
                      f"C restatement of the reference algorithm (not Julia), host has {os.cpu_count()} cores",
            "seconds": dt, "obj_gap_winner": gap, "oracle_opt_winner": float(objs[0])}


if __name__ == "__main__":
    main()
