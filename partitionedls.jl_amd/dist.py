"""Multi-GPU plumbing of the Opt sweep: shard the Gray-index space, then pick the global optimum with two tiny all-reduces.

One process per GPU; torch.distributed is only the transport (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  The reduction mirrors `argmin` at Opt.jl:96 — first minimal index on ties — as a lexicographic minimum over
(objective, pattern index): RCCL has no MINLOC, so the index all-reduce is masked to the ranks that hold the minimum.
"""
NO_CANDIDATE = (1 << 62)


def shard_range(npat, rank, world):
    """Gray-index range [g0, g1) of this rank: contiguous, disjoint, covering [0, npat)."""
    return rank * npat // world, (rank + 1) * npat // world


def allreduce_argmin(obj, pat, device=None, group=None):
    """(objective, pattern) of the global lexicographic minimum. `pat < 0` means this rank has no candidate."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return obj, pat
    o = torch.tensor([obj if pat >= 0 else float("inf")], dtype=torch.float64, device=device)
    dist.all_reduce(o, op=dist.ReduceOp.MIN, group=group)
    gmin = float(o[0])
    i = torch.tensor([pat if (pat >= 0 and obj == gmin) else NO_CANDIDATE], dtype=torch.int64, device=device)
    dist.all_reduce(i, op=dist.ReduceOp.MIN, group=group)
    gpat = int(i[0])
    return gmin, (gpat if gpat != NO_CANDIDATE else -1)
