#!/usr/bin/env python3
"""bench.py — the hot path of PartitionedLS on MI355X, one BASELINE.json config per run (default C3, the headline).

    python bench.py --gpus 1 --steps K --warmup W [--config C2|C3|C4|C5|L340]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over the synthetic problem, inputs resident in HBM when the timed region starts.
  C2 / C3 (fit(Opt)): Gram build (fp64 MFMA) -> tableau prep -> [long enumerations: bit-order calibration, ~0.4 ms, every step]
        -> sign-pattern sweep over this rank's shard of the Gray-index space
        -> all-reduce(min objective, then min index among the minimisers) over RCCL -> winner re-solve + objective from the data.
        The 2^K' patterns of ONE problem are sharded across ranks (total work fixed => "strong" scaling); no data-path collective
        besides the two 8-byte all-reduces.  value = patterns solved by all ranks / max-over-ranks time.
  C5 (fit(BnB), K = 24): the same Opt enumeration of 2^24 patterns (value, as above) followed by fit(BnB) with the frontier
        batches sharded across the ranks (dist.bnb_search: one all-gather of (bound, branch) per batch); BnB time and node
        count ride in `bnb`.  `bnb_hard` (rank 0, once, outside the timed region): the same search on a pure-noise target,
        where the relaxation is never feasible at the root, capped at --bnb-cap nodes: nodes bounded per second.
  C4 (fit(Alt), N = 1M, D = 512): Gram build + the ALS loop (T = 200, eps = 1e-6 as Alt.jl:50-51; it converges long before T).
        One GPU per north_star: with N ranks every rank fits its own replica ("weak": replicas only, no collective).
        value = fits/s over all ranks; the roofline object is the Gram kernel's fp64-MFMA fraction.
Rank 0 prints ONE JSON line (DESIGN.md §6 defines `roofline` (the binding fp64 bound), `roofline_hbm` (measured bytes),
`nominal_hbm` (SURVEY §8(d)'s byte model) and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (kind, seed, N, D, K)   — BASELINE.md §4
    "C2": ("opt", 20260002, 10_000, 128, 12),
    "C3": ("opt", 20260003, 100_000, 256, 20),
    "C4": ("alt", 20260004, 1_000_000, 512, 16),
    "C5": ("bnb", 20260005, 100_000, 256, 24),
    # not a BASELINE config: fit(Opt) beyond the register kernel (n = 340 > 320 tableau variables), the deferred-update kernel
    # sweep_lazy.hip; same problem as tools/generic_timing.py 20000 340 18, whose rocprofv3 passes give the HBM traffic below
    "L340": ("opt", 7, 20_000, 340, 18),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # public datasheet (vector = matrix fp64 on MI355X); not in the on-image guides
METRIC = "sign-pattern NNLS solves/sec (whole node); fp64 obj gap vs ref"


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
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg of C3 / C4")
    ap.add_argument("--cpu-patterns", type=int, default=32, help="patterns in the dense CPU baseline sample")
    ap.add_argument("--cpu-workers", type=int, default=0, help="worker processes of the CPU baselines (0 = min(16, cores))")
    ap.add_argument("--bnb-cap", type=int, default=200000, help="node cap of the C5 bnb_hard leg")
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

    kind, seed, N, D, K = CONFIGS[args.config]
    P, wstar = pls.synth_truth(seed, D, K)
    dX = torch.empty(N * D, dtype=torch.float64, device=dev)         # column-major N x D, resident in HBM
    dy = torch.empty(N, dtype=torch.float64, device=dev)
    ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
    torch.cuda.synchronize()
    flags = L.OPT_FAITHFUL_INTERCEPT if args.faithful else 0
    rng0 = np.random.default_rng(123)
    alt_a0, alt_b0 = rng0.random(D + 1), (rng0.random(K + 1) - 0.5) * 10    # the start Alt.jl:65-66 would draw

    def opt_pass():
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, flags)
        npat = ctx.num_patterns()
        g0, g1 = pls.dist.shard_range(npat, rank, world)
        # the ranges partition the pattern space only if every rank visits it in the same order: the key of this rank's order
        # (measured group -> Gray-bit assignment, deterministic in the data) rides in the first all-reduce and is compared there
        okey = pls.dist.order_key(ctx.bit_order()[0]) if world > 1 else None
        bobj, bpat, _, unconv = ctx.opt_sweep(g0, g1)
        t_gram, t_prep, t_sweep, t_calib = ctx.timing(L.T_GRAM), ctx.timing(L.T_PREP), ctx.timing(L.T_SWEEP), ctx.timing(L.T_CALIB)
        pivots, vetoes = ctx.pivots(), ctx.vetoes()
        # all-reduce(min residual), then one all-gather of every shard's winner + near ties (64 B per rank): the first-index argmin of
        # Opt.jl:96 and the candidate set a single context would re-rank on the data objective — every rank finishes the same model
        _, bpat = pls.dist.reduce_winner(ctx, bobj, bpat, device=red_dev, order_key=okey)
        a, b, t, opt, bi = ctx.opt_finish(bpat)
        return dict(npat=npat, local=g1 - g0, opt=opt, best_index=bi, unconv=unconv, t_gram=t_gram, t_prep=t_prep,
                    t_sweep=t_sweep, t_calib=t_calib, t_finish=ctx.timing(L.T_FINISH), pivots=pivots, vetoes=vetoes)

    def bnb_pass(cap=None):
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
        t0 = time.perf_counter()
        if world == 1:          # in-library search: every node warm-started from its parent's tableau snapshot (partls_bnb_search)
            mu, pat, free, bounded = ctx.bnb_search(cap or 0)
        else:                   # frontier batches dealt over the ranks, every node to the rank that holds its parent's tableau snapshot
            mu, pat, free, bounded = pls.dist.bnb_search_warm(ctx, K + 1, rank=rank, world=world, device=red_dev, max_nodes=cap)
        a, b, t, opt = ctx.bnb_leaf(pat, free)
        return dict(opt=opt, nopen=bounded, seconds=time.perf_counter() - t0)

    def alt_pass():
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
        t_gram, t_prep = ctx.timing(L.T_GRAM), ctx.timing(L.T_PREP)
        t0 = time.perf_counter()
        a, b, t, opt, iters = ctx.alt_prepared(alt_a0, alt_b0, eps=1e-6, T=200)
        return dict(opt=opt, iters=iters, t_gram=t_gram, t_prep=t_prep, alt_ms=(time.perf_counter() - t0) * 1e3)

    def step():
        if kind == "alt":
            return alt_pass()
        r = opt_pass()
        if kind == "bnb":
            r["bnb"] = bnb_pass()
        return r

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    hist = []
    for _ in range(args.steps):
        hist.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    res = hist[-1]
    avg = lambda key: sum(h[key] for h in hist) / len(hist)
    Mp = D + 1
    out = {"metric": METRIC, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "vs_baseline": None, "dtype": "f64",
           "data": "synthetic"}
    gram_flops = 2.0 * N * (D + 2) ** 2 / 2                      # useful flops of the SYRK-shaped Gram build (upper triangle)

    if kind in ("opt", "bnb"):
        npat = res["npat"]
        out["value"] = npat * args.steps / elapsed
        out["scaling"] = "strong"
        out["config"] = {"workload": f"{args.config}: fit({'Opt' if kind == 'opt' else 'Opt enumeration + BnB'}) N={N} D={D} K={K}, "
                                     f"{npat} sign patterns ({'2^(K+1) faithful' if args.faithful else '2^K, free intercept'}), eta=0",
                         "seed": seed, "sharding": f"gray-index range / {world} ranks",
                         "enumeration": "every one of the sign patterns is solved to KKT optimality in every step; they are visited along Gray-code "
                                        "chains with the groups assigned to the Gray bits by their measured flip cost (calibration inside "
                                        "every step: kernels_ms.bit_order_calibration; 0 = the enumeration is too short to repay it)"}
        # dominant kernel = the sweep; its launch processes this rank's shard; duration from HIP events on the library's stream
        sweep_avg_s = avg("t_sweep") * 1e-3
        solves_per_launch = res["local"]
        bytes_per_solve = algorithmic_bytes_per_solve(Mp)
        achieved_gbs = solves_per_launch * bytes_per_solve / sweep_avg_s / 1e9
        # fp64 view of the same kernel: every pivot is one rank-1 update of the symmetric tableau = T(T+1)/2 tile slots x 256 FMAs
        n_tab = Mp if args.faithful else D
        tiles = (n_tab + 15) // 16
        flops = res["pivots"] * (tiles * (tiles + 1) // 2) * 256 * 2
        fp64_tflops = flops / sweep_avg_s / 1e12
        # HBM traffic of one sweep launch: PMC counters cannot be read inside this process, so the per-launch figure is the one the
        # round's rocprofv3 FETCH_SIZE / WRITE_SIZE passes of THIS command measured (tools/profile_r0N.sh -> profiles/)
        traffic, tsrc = None, None
        for tname in ("r04_sweep_traffic.json", "r03_sweep_traffic.json", "r02_sweep_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if world == 1 and args.config == "C3" and not args.faithful and os.path.exists(tpath):
                traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
                tsrc = "offline PMC (rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this kernel, corrected per MI355X_MICROARCH.md), " \
                       f"profiles/{tname} — not measured in this run"
                break
        # The binding bound of the sweep is fp64 FMA issue: the tableau lives in registers and is never re-read from HBM, so every
        # pivot costs one rank-1 update of the symmetric tableau = T(T+1)/2 tile slots x 256 FMAs, whatever the memory system does.
        if n_tab > 320:
            # Beyond the register kernel the tableau lives in global memory (sweep_lazy.hip): the bound that binds is HBM.  Bytes per
            # sweep launch from the round's rocprofv3 FETCH_SIZE / WRITE_SIZE passes of the same problem (tools/profile_r03.sh).
            lz_traffic, lz_src = None, None
            tpath = os.path.join(ROOT, "profiles", "r04_d340_traffic.json")
            if not os.path.exists(tpath): tpath = os.path.join(ROOT, "profiles", "r03_d340_traffic.json")
            if world == 1 and args.config == "L340" and not args.faithful and os.path.exists(tpath):
                lz_traffic = json.load(open(tpath))["deferred_update_kernel"].get("hbm_bytes_per_sweep")
                lz_src = "offline PMC (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of this kernel on this problem), profiles/" + os.path.basename(tpath) + " — not measured in this run"
            pass_flops = res["pivots"] * float(n_tab) * n_tab          # every pivot: one rank-1 update of the triangle, n^2/2 entries x 2 flop
            out["roofline"] = {"bound": "hbm", "achieved": None if lz_traffic is None else lz_traffic / sweep_avg_s / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": None if lz_traffic is None else lz_traffic / sweep_avg_s / 1e9 / HBM_PEAK_GBS,
                               "traffic": lz_traffic, "traffic_source": lz_src, "kernel": "sweep_lazy_kernel", "kernel_ms": sweep_avg_s * 1e3,
                               "pivots_per_launch": res["pivots"], "solves_per_launch": solves_per_launch,
                               "note": "MEASURED bytes (the kernel's own traffic: passes over the base image, gathered and replaced rows / "
                                       "columns) / HIP-event time of the sweep launch; the eager kernel of round 2 moves 5x more"}
            out["roofline_fp64"] = {"bound": "fp64-mfma", "achieved": pass_flops / sweep_avg_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": pass_flops / sweep_avg_s / 1e12 / FP64_PEAK_TFLOPS,
                                    "note": "pivots x n^2 flop (the rank-R passes over the base image, v_mfma_f64_16x16x4) / sweep time"}
            out["kernels_ms"] = {"gram_build": avg("t_gram"), "prep": res["t_prep"], "bit_order_calibration": avg("t_calib"),
                                 "sweep": sweep_avg_s * 1e3, "finish": res["t_finish"]}
            out["result"] = {"opt": res["opt"], "best_index": res["best_index"], "unconverged": res["unconv"], "loo_vetoes": res["vetoes"]}
        else:
            out["roofline"] = {"bound": "fp64-valu", "achieved": fp64_tflops,   "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": fp64_tflops / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                               "kernel": "sweep_blk_kernel", "kernel_ms": sweep_avg_s * 1e3, "pivots_per_launch": res["pivots"],
                               "flop_per_pivot": (tiles * (tiles + 1) // 2) * 256 * 2, "solves_per_launch": solves_per_launch,
                               "note": "algorithmic flops = principal pivots x tile slots x 256 x 2 (counted by the kernel) / HIP-event time of "
                                       "the sweep launch; peak = 78.6 TFLOP/s fp64 vector (public datasheet; not in the on-image guide)"}
            out["roofline_hbm"] = None if traffic is None else {
                "bound": "hbm", "achieved": traffic / sweep_avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": traffic / sweep_avg_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": traffic,
                "note": "MEASURED HBM bytes of the sweep launch (chain-start tableau loads and their scratch) / kernel time: HBM is idle"}
            out["nominal_hbm"] = {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_solve": bytes_per_solve,
                                  "note": "SURVEY §8(d) nominal bytes (G re-read per solve) x solves / kernel time. NOT a roofline fraction: "
                                          "warm-started Gray chains keep the tableau in registers, the kernel never moves these bytes, so "
                                          "the figure can exceed 1"}
            out["kernels_ms"] = {"gram_build": avg("t_gram"), "prep": res["t_prep"], "bit_order_calibration": avg("t_calib"),
                                 "sweep": sweep_avg_s * 1e3, "finish": res["t_finish"]}
            out["gram"] = {"tflops_useful": gram_flops / (avg("t_gram") * 1e-3) / 1e12, "frac_of_fp64_mfma_peak":
                           gram_flops / (avg("t_gram") * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
            out["result"] = {"opt": res["opt"], "best_index": res["best_index"], "unconverged": res["unconv"], "loo_vetoes": res["vetoes"]}
        if kind == "bnb":
            out["bnb"] = {"seconds": sum(h["bnb"]["seconds"] for h in hist) / len(hist), "nodes_bounded": res["bnb"]["nopen"],
                          "opt": res["bnb"]["opt"], "gap_vs_opt": abs(res["bnb"]["opt"] - res["opt"]) / res["opt"],
                          "sharding": ("in-library search, warm-started node bounds" if world == 1 else
                                       f"frontier batches of 1024 x {world} nodes, each node bounded by the rank that holds its parent's tableau "
                                       f"snapshot (warm start), one all-gather per batch")}
    else:
        out["metric"] = "fit(Alt) fits/sec (Gram build + ALS loop); Gram build fp64-MFMA fraction"
        out["unit"] = "fits/s"
        out["value"] = world * args.steps / elapsed
        out["scaling"] = "weak"
        out["config"] = {"workload": f"{args.config}: fit(Alt) N={N} D={D} K={K}, T=200, eps=1e-6, start drawn as Alt.jl:65-66 "
                                     f"(numpy seed 123), eta=0", "seed": seed,
                         "sharding": "replicas only: one GPU per fit (north_star), no collective"}
        gram_s = avg("t_gram") * 1e-3
        tf = gram_flops / gram_s / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                           "traffic": None, "traffic_source": None, "kernel": "gram_kernel", "kernel_ms": gram_s * 1e3,
                           "algorithmic_flops": gram_flops, "note": "useful flops of the upper triangle of [X 1 y]'[X 1 y]"}
        out["kernels_ms"] = {"gram_build": gram_s * 1e3, "prep": res["t_prep"], "alt_loop": avg("alt_ms")}
        out["alt"] = {"iterations": res["iters"], "ms_per_iteration": avg("alt_ms") / max(1, res["iters"]), "opt": res["opt"],
                      "noise_floor": 0.1 * N ** 0.5}

    if rank == 0 and world == 1 and args.config in ("C3", "C4") and not args.no_host_inclusive:
        # What a caller with X in HOST memory sees (the reference's callers hold Julia arrays, Opt.jl:73 / Alt.jl:50): the same fit through
        # the host-pointer entry, X uploaded inside the call.  Never `value`.  Two cases: a FRESH array (pages the runtime has never
        # pinned: a new allocation per fit) and the same array again.
        hX = dX.view(D, N).t().cpu().numpy()                         # column-major on the device -> an F-ordered (N, D) host view
        hX = np.asfortranarray(hX); hy = dy.cpu().numpy()

        def host_fit(Xh):
            t1 = time.perf_counter()
            if kind == "alt":
                ctx.opt_prepare(Xh, hy, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
                up = ctx.upload()
                r = ctx.alt_prepared(alt_a0, alt_b0, eps=1e-6, T=200)[3]
            else:
                ctx.opt_prepare(Xh, hy, P, 0.0, flags)
                up = ctx.upload()
                bo, bp, _, _ = ctx.opt_sweep(0, -1)
                r = ctx.opt_finish(bp)[3]
            return (time.perf_counter() - t1) * 1e3, up, r
        fresh = []
        for _ in range(2):
            Xf = np.empty_like(hX, order="F"); Xf[...] = hX              # a new allocation: first touch by the copier threads / the runtime
            fresh.append(host_fit(Xf)); del Xf
        again = [host_fit(hX) for _ in range(3)]
        best = min(again, key=lambda v: v[0])
        out["host_inclusive"] = {
            "ms_per_fit": best[0], "ms_per_fit_fresh_array": min(f[0] for f in fresh), "upload_ms": best[1][0], "upload_bytes": best[1][1],
            "pcie_gbs": best[1][1] / best[1][0] / 1e6, "pcie_gbs_fresh_array": max(f[1][1] / f[1][0] / 1e6 for f in fresh),
            "device_resident_ms": out["ms_per_step"], "opt": best[2],
            "note": "the same fit through the host-pointer entry (X uploaded inside the call, staged through page-locked buffers by 4 copier "
                    "threads): what a Julia caller sees; the link's ceiling on this box is 57 GB/s (tools/ubench/h2d_paths.hip).  Not `value`."}
    if rank == 0 and kind == "bnb":
        # BnB where it has to branch: target = intercept + noise (no feature carries signal), same X, same shape
        ctx.synth_device(seed, N, D, np.zeros(D), dX.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
        t1 = time.perf_counter()
        mu, pat, free, bounded = ctx.bnb_search(args.bnb_cap)
        dt_first = time.perf_counter() - t1                  # the context's first long search also grows the snapshot pool (hipMalloc of
        t1 = time.perf_counter()                             # 160 MB chunks, kept by the context): reported separately
        mu, pat, free, bounded = ctx.bnb_search(args.bnb_cap)
        dt = time.perf_counter() - t1
        out["bnb_hard"] = {"target": "y = 1 + 0.1 * noise (wstar = 0)", "nodes_bounded": bounded, "seconds": dt,
                           "nodes_per_s": bounded / dt, "first_search_seconds": dt_first, "first_search_nodes_per_s": bounded / dt_first,
                           "capped": bounded >= args.bnb_cap, "incumbent": (mu if mu != float("inf") else None),
                           "search": "partls_bnb_search: best-first, device batches, children warm-started from the parent's tableau snapshot; "
                                     "`seconds` = the second search on this context (snapshot pool already allocated), `first_search_*` = "
                                     "the first one, which pays the pool's hipMalloc chunks"}
        if bounded < args.bnb_cap:
            # certify the search: the full 2^K enumeration of the SAME problem must find the same optimum (BnB.jl:94-132 vs Opt.jl:85-96)
            ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, 0)
            bo_h, bp_h, _, unc_h = ctx.opt_sweep(0, -1)
            opt_h = ctx.opt_finish(bp_h)[3]
            leaf_h = ctx_leaf_opt(ctx, dX, dy, N, D, P, pat, free, L)
            out["bnb_hard"].update({"opt_enumeration": opt_h, "bnb_leaf_opt": leaf_h, "gap_vs_opt": abs(leaf_h - opt_h) / opt_h,
                                    "gap_incumbent_vs_opt": abs(mu - opt_h) / opt_h, "enumeration_unconverged": unc_h})
            ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
        t1 = time.perf_counter()                             # the same frontier logic with every node from the fresh tableau
        mu2, _, _, bounded2 = pls.dist.bnb_search(ctx.bnb_bound, K + 1, max_nodes=args.bnb_cap)
        dt2 = time.perf_counter() - t1
        out["bnb_hard"]["cold_nodes_per_s"] = bounded2 / dt2
        t1 = time.perf_counter()                             # the host-driven form of the warm search (what N > 1 ranks run): native frontier,
        mu3, _, _, bounded3 = pls.dist.bnb_search_warm(ctx, K + 1, max_nodes=args.bnb_cap)     # snapshot slots through the C ABI, one rank
        out["bnb_hard"]["host_driven_warm_nodes_per_s"] = bounded3 / (time.perf_counter() - t1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and kind != "alt":
        # the sample of the CPU baseline: the GPU's winner + random patterns (reference indexing, K + 1 bits).  The device's answer for
        # every one of them — from the SWEEP's own per-pattern output (all_opt of a faithful enumeration: what ranks the patterns) and
        # from a single re-solve with the objective recomputed from the data (partls_opt_pattern) — is compared with the dense oracle:
        # SURVEY §8(d) "per sampled pattern and for the global optimum"
        npat_ref = 1 << (K + 1)
        n = max(1, args.cpu_patterns)
        sample = [int(res["best_index"])] + [int(v) for v in np.random.default_rng(0).integers(0, npat_ref, size=n - 1)]
        if kind == "bnb":                                        # the BnB leg re-generated y: restore the config's problem
            ctx.synth_device(seed, N, D, wstar, dX.data_ptr(), dy.data_ptr())
            torch.cuda.synchronize()
        ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
        _, _, allopt, unc = ctx.opt_sweep(0, -1, want_all=True)
        dev_sweep = [float(allopt[b]) for b in sample]
        del allopt
        dev_solve = [ctx.opt_pattern(b)[1] for b in sample]
        out["cpu_baseline"] = cpu_baseline(args, seed, N, D, K, res, sample, dev_sweep, dev_solve, np)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def ctx_leaf_opt(ctx, dX, dy, N, D, P, pat, free, L):
    """objective (from the data) of the model of one BnB node"""
    ctx.opt_prepare_device(dX.data_ptr(), dy.data_ptr(), N, D, N, P, 0.0, L.OPT_FAITHFUL_INTERCEPT)
    return ctx.bnb_leaf(pat, free)[3]


def cpu_baseline(args, seed, N, D, K, res, sample, dev_sweep, dev_solve, np):
    """A: the oracle in reference-faithful dense mode (Opt.jl:87-90 per pattern: column scaling + dense Lawson–Hanson on the
    N x (D+1) matrix + objective) on a bounded sample that always includes the GPU's winner; every sampled objective is compared
    with the device's (the fp64 objective gap).  B: the same NNLS on the QR-compressed problem — the CPU analogue of the Gram
    form — on 1 core and on all workers.  Both run in spawned worker processes of oracle/cpu_baseline.py."""
    from oracle import cpu_baseline as CB
    workers = args.cpu_workers or CB.default_workers()
    npat_ref = 1 << (K + 1)
    n = len(sample)
    objs, wall, cpu_s, setup = CB.dense_sample(seed, N, D, K, sample, min(workers, n))
    rel = lambda a, b: abs(a - b) / max(1.0, b)
    gap = rel(res["opt"], objs[0])
    gap_sweep = max(rel(a, b) for a, b in zip(dev_sweep, objs))
    gap_solve = max(rel(a, b) for a, b in zip(dev_solve, objs))
    b1, bn, bsetup = CB.compressed_rates(seed, N, D, K, 64 if N * D > 5_000_000 else 256, workers)
    return {"value": n / wall, "unit": "solves/s", "cores": min(workers, n), "kind": "port",
            "sample": f"{n} of {npat_ref} patterns (winner + random), dense Lawson-Hanson per pattern as Opt.jl:87-90, "
                      f"C restatement of the reference algorithm (not Julia), one pattern per worker call; host has {os.cpu_count()} cores",
            "seconds": wall, "per_core_solves_per_s": n / cpu_s, "obj_gap_winner": gap, "oracle_opt_winner": float(objs[0]),
            "obj_gap_sampled_max": gap_sweep, "obj_gap_sampled_max_resolved": gap_solve,
            "obj_gap_note": "|device - oracle| / max(1, oracle) over the whole sample; `sampled_max`: the sweep's own per-pattern "
                            "objective (all_opt of a faithful 2^(K+1) enumeration, Gram form); `resolved`: partls_opt_pattern "
                            "(single solve, objective from the data); `winner`: the timed run's opt",
            "gram_form": {"what": "Baseline B (BASELINE.md §3.2): the same NNLS on the QR-compressed (D+2) x (D+1) problem, i.e. the "
                                  "data are read once as on the GPU; compression time not included",
                          "solves_per_s_1_core": b1, "solves_per_s_all_workers": bn, "workers": workers,
                          "compress_seconds_1_core": bsetup}}


if __name__ == "__main__":
    main()
