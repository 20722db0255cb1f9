"""Multi-GPU plumbing: the Opt sweep shards the Gray-index space and picks the global optimum with two tiny all-reduces; the BnB
search shards every frontier batch and shares bounds (hence the incumbent) with one all-gather per batch.

One process per GPU; torch.distributed is only the transport (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  The reduction mirrors `argmin` at Opt.jl:96 — first minimal index on ties — as a lexicographic minimum over
(objective, pattern index): RCCL has no MINLOC, so the index all-reduce is masked to the ranks that hold the minimum.
"""
NO_CANDIDATE = (1 << 62)


def shard_range(npat, rank, world):
    """Gray-index range [g0, g1) of this rank: contiguous, disjoint, covering [0, npat)."""
    return rank * npat // world, (rank + 1) * npat // world


def order_key(gbit):
    """A number every rank of a sharded sweep must agree on: the Gray-index ranges of shard_range only partition the pattern space
    when all ranks visit it in the same order (Context.bit_order).  Exact in float64 (it rides in the objective all-reduce)."""
    import zlib
    import numpy as np
    return float(zlib.crc32(np.asarray(gbit, dtype=np.int64).tobytes()))


def allreduce_argmin(obj, pat, device=None, group=None, order_key=None):
    """(objective, pattern) of the global lexicographic minimum. `pat < 0` means this rank has no candidate.
    order_key (optional, see order_key()): checked for equality across the ranks inside the same all-reduce."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return obj, pat
    key = 0.0 if order_key is None else float(order_key)
    o = torch.tensor([obj if pat >= 0 else float("inf"), key, -key], dtype=torch.float64, device=device)
    dist.all_reduce(o, op=dist.ReduceOp.MIN, group=group)
    o = o.cpu()
    if float(o[1]) != -float(o[2]):
        raise RuntimeError("sharded Opt sweep: the ranks visit the patterns in different orders (Context.bit_order differs), "
                           "so their Gray-index ranges do not partition the pattern space; prepare the same problem on every rank")
    gmin = float(o[0])
    i = torch.tensor([pat if (pat >= 0 and obj == gmin) else NO_CANDIDATE], dtype=torch.int64, device=device)
    dist.all_reduce(i, op=dist.ReduceOp.MIN, group=group)
    gpat = int(i[0])
    return gmin, (gpat if gpat != NO_CANDIDATE else -1)


def reduce_winner(ctx, obj, pat, device=None, group=None, order_key=None):
    """The sharded fit(Opt)'s reduction, near ties included: (objective, pattern) of the global lexicographic minimum, installed on
    `ctx` together with the near ties of EVERY shard, so that the next ctx.opt_finish(pattern) re-ranks the candidate set a single
    context would have had (Opt.jl:90,96: objectives from the data, first index on ties) and every rank returns the same model.

    One all-reduce(min) on [objective, key, -key] — the min-residual reduction of the north star; the key of the visiting order is
    checked inside it — and one all-gather of every rank's (at most 4) candidates: 64 bytes per rank, which also carries the index
    (it replaces the masked index all-reduce of allreduce_argmin).  With one rank (or no process group) nothing is exchanged."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return obj, pat
    world = dist.get_world_size(group)
    key = 0.0 if order_key is None else float(order_key)
    o = torch.tensor([obj if pat >= 0 else float("inf"), key, -key], dtype=torch.float64, device=device)
    dist.all_reduce(o, op=dist.ReduceOp.MIN, group=group)
    o = o.cpu()
    if float(o[1]) != -float(o[2]):
        raise RuntimeError("sharded Opt sweep: the ranks visit the patterns in different orders (Context.bit_order differs), "
                           "so their Gray-index ranges do not partition the pattern space; prepare the same problem on every rank")
    co, cp = ctx.opt_candidates() if pat >= 0 else (np.zeros(0), np.zeros(0, dtype=np.int64))
    buf = torch.full((8,), float("inf"), dtype=torch.float64, device=device)       # [4 objectives | 4 patterns (exact in a double: < 2^41)]
    buf[4:] = -1.0
    if len(co):
        buf[:len(co)] = torch.as_tensor(co)
        buf[4:4 + len(cp)] = torch.as_tensor(cp.astype(np.float64))
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    g = torch.stack(out).cpu().numpy()
    gobj, gpat = ctx.opt_merge_candidates(g[:, :4].ravel(), g[:, 4:].ravel().astype(np.int64))
    if gpat >= 0 and gobj != float(o[0]):
        raise RuntimeError("sharded Opt sweep: the all-reduced minimum and the gathered candidates disagree")
    return gobj, gpat


def bnb_search(bound_fn, n_groups, rank=0, world=1, group=None, batch=512, device=None, max_nodes=None):
    """fit_BnB (BnB.jl:94-132) as a best-first search whose frontier batches are sharded across ranks.

    Every rank holds the SAME frontier (it is rebuilt from shared data only, so it never has to be exchanged): per round the
    `batch * world` most promising nodes are popped, rank r bounds the nodes r, r + world, ... on its own GPU with
    `bound_fn(pats, frees) -> (lb, branch)` (Context.bnb_bound), one all-gather shares the (bound, branch) pairs — which carries
    the incumbent, the min over the feasible bounds — and every rank applies the same pruning and branching.  Returns
    (mu, best_pat, best_free, nodes_bounded); the caller builds the model of that node locally (Context.bnb_leaf), as every
    rank re-solves the winner after the Opt sweep.  A node is (pat, free): bit k of free set = group k not branched yet.
    max_nodes (measurement only): stop after that many bounded nodes and return the incumbent so far (inf if none yet).
    """
    import heapq
    import numpy as np
    frontier = [(0.0, 0, 0, (1 << n_groups) - 1)]               # (parent bound, sequence number, pat, free): root = all free
    seq = 1
    mu, best = float("inf"), None
    bounded = 0
    use_dist = world > 1
    if use_dist:
        import torch
        import torch.distributed as dist
    while frontier:
        if max_nodes is not None and bounded >= max_nodes:
            break
        nodes = []
        while frontier and len(nodes) < batch * world:
            key, _, pat, free = heapq.heappop(frontier)
            if key >= mu:
                continue                                         # its bound can only be >= the parent's (BnB.jl:102)
            nodes.append((pat, free))
        if not nodes:
            break
        mine = nodes[rank::world]
        lb_l, br_l = bound_fn(np.array([p for p, _ in mine], dtype=np.uint64), np.array([f for _, f in mine], dtype=np.uint64))
        if use_dist:
            per = (len(nodes) + world - 1) // world              # equal-sized slots so that one all_gather fits every rank
            buf = torch.full((2 * per,), float("nan"), dtype=torch.float64, device=device)
            buf[:len(mine)] = torch.as_tensor(np.asarray(lb_l, dtype=np.float64))
            buf[per:per + len(mine)] = torch.as_tensor(np.asarray(br_l, dtype=np.float64))
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf, group=group)
            lb = np.empty(len(nodes)); br = np.empty(len(nodes), dtype=np.int64)
            for r in range(world):
                cnt = len(nodes[r::world])
                o = out[r].cpu().numpy()
                lb[r::world] = o[:cnt]
                br[r::world] = o[per:per + cnt].astype(np.int64)
        else:
            lb, br = np.asarray(lb_l, dtype=np.float64), np.asarray(br_l, dtype=np.int64)
        for (pat, free), l, k in zip(nodes, lb, br):
            bounded += 1
            if l >= mu:
                continue
            if k < 0:                                            # feasible for the original problem (BnB.jl:109-115)
                mu, best = float(l), (pat, free)
                continue
            bit = 1 << int(k)
            heapq.heappush(frontier, (float(l), seq, pat | bit, free & ~bit)); seq += 1     # alpha_pk >= 0 first (BnB.jl:120,123)
            heapq.heappush(frontier, (float(l), seq, pat & ~bit, free & ~bit)); seq += 1    # alpha_pk <= 0
    if best is None:
        if max_nodes is not None:
            return mu, 0, (1 << n_groups) - 1, bounded
        raise RuntimeError("bnb_search: no feasible leaf found")
    return mu, best[0], best[1], bounded


def bnb_search_warm(ctx, n_groups, rank=0, world=1, group=None, batch=1024, device=None, max_nodes=None):
    """The rank-sharded best-first search of bnb_search with WARM-STARTED node bounds and a NATIVE frontier: a node is bounded on the
    rank that holds its parent's tableau snapshot (Context.bnb_bound_snap), so a child exchanges only the variables of the one group
    that was branched instead of being solved from the fresh tableau (4-5x fewer device cycles per node), and all
    per-node bookkeeping — popping, pruning, dealing, branching, snapshot reference counts — runs in the library
    (include/partls.h: partls_frontier_*), a few calls per ROUND of `batch * world` nodes; Python only carries the one all-gather of
    (bound, branch, new slot) per round.

    Every rank runs the same frontier on the same data.  Dealing: a node with a snapshot goes to its owner, up to the owner's quota
    of the round; what exceeds it, and every node without a snapshot, goes to the least loaded ranks and starts COLD there (the work
    spreads over the ranks by itself: the root's children are owned by rank 0, their surplus seeds subtrees elsewhere).
    Returns (mu, best_pat, best_free, nodes_bounded) like bnb_search."""
    import numpy as np
    from .api import Frontier
    use_dist = world > 1
    if use_dist:
        import torch
        import torch.distributed as dist
    ctx.bnb_snap_begin()
    fr = Frontier(n_groups, rank, world, batch)
    try:
        while True:
            if max_nodes is not None and fr.result()[3] >= max_nodes:
                break
            total, pats, frees, srcs, per_rank = fr.next()
            if total == 0:
                ctx.bnb_snap_release(fr.ingest(np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)))
                break
            lb_l, br_l, dst_l = ctx.bnb_bound_snap(pats, frees, srcs)
            if use_dist:
                per = int(per_rank.max())                        # equal-sized slots so that one all_gather fits every rank
                mine = len(pats)
                buf = torch.full((3 * per,), float("nan"), dtype=torch.float64, device=device)
                buf[:mine] = torch.as_tensor(np.asarray(lb_l, dtype=np.float64))
                buf[per:per + mine] = torch.as_tensor(np.asarray(br_l, dtype=np.float64))
                buf[2 * per:2 * per + mine] = torch.as_tensor(np.asarray(dst_l, dtype=np.float64))
                out = [torch.empty_like(buf) for _ in range(world)]
                dist.all_gather(out, buf, group=group)
                o = [t.cpu().numpy() for t in out]
                lb = np.concatenate([o[r][:per_rank[r]] for r in range(world)])
                br = np.concatenate([o[r][per:per + per_rank[r]] for r in range(world)]).astype(np.int32)
                dst = np.concatenate([o[r][2 * per:2 * per + per_rank[r]] for r in range(world)]).astype(np.int32)
            else:
                lb, br, dst = lb_l, br_l, dst_l
            ctx.bnb_snap_release(fr.ingest(lb, br, dst))
        mu, pat, free, bounded = fr.result()
    finally:
        fr.close()
    if not (mu < float("inf")):
        if max_nodes is not None:
            return mu, 0, (1 << n_groups) - 1, bounded
        raise RuntimeError("bnb_search_warm: no feasible leaf found")
    return mu, pat, free, bounded
