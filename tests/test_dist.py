"""CPU tests of the multi-rank path (gloo, world_size 2): shard ranges partition the pattern space and the two-step
all-reduce returns the lexicographic (objective, index) minimum, i.e. argmin with first-index ties (Opt.jl:96)."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cases, out):
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = []
    for case in cases:
        obj, pat = case[rank]
        res.append(pls.dist.allreduce_argmin(obj, pat))
    out[rank] = res
    dist.destroy_process_group()


def test_shard_ranges_partition_the_pattern_space(partls):
    for npat in [1, 8, 1 << 13, (1 << 20) + 3]:
        for world in [1, 2, 3, 4, 8]:
            r = [partls.dist.shard_range(npat, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == npat
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))


def test_allreduce_argmin_gloo_world2(partls):
    import torch.multiprocessing as mp
    cases = [
        [(3.0, 10), (2.0, 99)],            # rank 1 wins on objective
        [(2.0, 77), (2.0, 5)],             # tie on objective: smaller pattern index wins (first-index argmin)
        [(1.5, 4), (float("inf"), -1)],    # rank 1 has no candidate (empty shard)
        [(0.0, 1 << 40), (0.0, (1 << 40) + 1)],
    ]
    expect = [(2.0, 99), (2.0, 5), (1.5, 4), (0.0, 1 << 40)]
    port = _free_port()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cases, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert [tuple(x) for x in out[r]] == expect


def _order_worker(rank, world, port, gbits, out):
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = []
    for per_rank in gbits:
        try:
            res.append(pls.dist.allreduce_argmin(1.0 + rank, 7 + rank, order_key=pls.dist.order_key(per_rank[rank])))
        except RuntimeError as e:
            res.append("mismatch" if "different orders" in str(e) else repr(e))
    out[rank] = res
    dist.destroy_process_group()


def test_sharded_sweep_refuses_ranks_with_different_visiting_orders(partls):
    """Gray-index ranges partition the pattern space only when every rank visits it in the same order (the group -> Gray-bit
    assignment a context calibrates): the key of the order rides in the objective all-reduce and a mismatch raises on every rank."""
    import torch.multiprocessing as mp
    same = [3, 0, 2, 1]
    gbits = [[same, same], [same, [3, 0, 1, 2]]]
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, gbits, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert tuple(out[r][0]) == (1.0, 7) and out[r][1] == "mismatch"
    assert partls.dist.order_key(same) == partls.dist.order_key(list(same)) != partls.dist.order_key([0, 1, 2, 3])


# ---- BnB: frontier batches sharded across ranks (dist.bnb_search); the bound function here is the CPU oracle's NNLS -----------
def _bnb_problem():
    import numpy as np
    rng = np.random.default_rng(77)
    N, D, K = 60, 12, 4
    X = rng.standard_normal((N, D))
    y = rng.standard_normal(N)                                  # no signal: the relaxation is rarely feasible, the search branches
    P = np.zeros((D, K), dtype=np.int64)
    P[np.arange(D), np.arange(D) % K] = 1
    return X, y, P


def _oracle_bound_fn(X, y, P):
    """(pats, frees) -> (lb, branch): BnB.jl:69-92,107,117 restated with the oracle's dense NNLS on [Xp Xm]."""
    import numpy as np
    from oracle import oracle as O
    N, M = X.shape
    K = P.shape[1]
    Xo = np.column_stack([X, np.ones(N)])
    groups = [np.nonzero(P[:, k])[0] for k in range(K)] + [np.array([M])]

    def bound(pats, frees):
        lbs, brs = [], []
        for pat, free in zip(pats.tolist(), frees.tolist()):
            cols, owner, sign = [], [], []
            state = np.zeros(M + 1, dtype=int)                    # bit 0: alpha >= 0 present, bit 1: alpha <= 0 present
            for k, g in enumerate(groups):
                if not (free >> k) & 1:
                    state[g] |= 1 if (pat >> k) & 1 else 2
            for m in range(M + 1):
                if not state[m] & 2:
                    cols.append(Xo[:, m]); owner.append(m); sign.append(1.0)
                if not state[m] & 1:
                    cols.append(-Xo[:, m]); owner.append(m); sign.append(-1.0)
            A = np.column_stack(cols) if cols else np.zeros((N, 1))
            x, rn, mode, _ = O.nnls(A, y)
            w = np.zeros(M + 1)
            for xi, m, sg in zip(x, owner, sign):
                w[m] += sg * xi
            nu = [float(np.clip(w[g], 0, None).sum() * np.clip(-w[g], 0, None).sum()) if (free >> k) & 1 else 0.0
                  for k, g in enumerate(groups)]
            kb = int(np.argmax(nu))
            lbs.append(rn); brs.append(kb if nu[kb] > 0 else -1)
        return np.array(lbs), np.array(brs, dtype=np.int32)
    return bound


def _bnb_worker(rank, world, port, out):
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    X, y, P = _bnb_problem()
    out[rank] = pls.dist.bnb_search(_oracle_bound_fn(X, y, P), P.shape[1] + 1, rank=rank, world=world, batch=4)
    dist.destroy_process_group()


def test_bnb_search_sharded_gloo_world2(partls, oracle):
    """2 ranks (frontier batches dealt round-robin, one all-gather per batch) == 1 rank == the oracle's depth-first fit_BnB."""
    import torch.multiprocessing as mp
    X, y, P = _bnb_problem()
    single = partls.dist.bnb_search(_oracle_bound_fn(X, y, P), P.shape[1] + 1, batch=4)
    ref = oracle.fit_bnb(X, y, P)
    assert abs(single[0] - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"])
    assert single[3] > 8                                         # the instance really branches
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_bnb_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(2):
        mu, pat, free, bounded = out[r]
        assert (pat, free) == (single[1], single[2]) and abs(mu - single[0]) <= 1e-12 * max(1.0, single[0])
    assert out[0][3] == out[1][3]                                # replicated frontier: identical node counts on both ranks


class _FakeSnapCtx:
    """Stands in for Context in bnb_search_warm on the CPU: bounds come from the oracle; the snapshot pool is bookkeeping only, which
    is what the test checks — every slot handed out is returned exactly once, nobody starts from a slot that is not live."""

    def __init__(self, X, y, P, capacity=10_000):
        self._bound = _oracle_bound_fn(X, y, P)
        self.capacity = capacity
        self.live = set()
        self.next = 0
        self.handed = 0
        self.warm = 0
        self.cold = 0

    def bnb_snap_begin(self):
        self.live.clear()

    def bnb_bound_snap(self, pats, frees, srcs):
        import numpy as np
        lb, br = self._bound(pats, frees)
        dst = np.full(len(pats), -1, dtype=np.int32)
        for i, (f, s) in enumerate(zip(frees.tolist(), srcs.tolist())):
            assert s == -1 or s in self.live, "a node was started from a snapshot that is not live"
            self.warm += s >= 0
            self.cold += s < 0
            if f and len(self.live) < self.capacity:
                dst[i] = self.next; self.live.add(self.next); self.next += 1; self.handed += 1
        return lb, br, dst

    def bnb_snap_release(self, slots):
        for s in list(slots):
            assert s in self.live, "a slot was released twice (or never handed out)"
            self.live.discard(s)


def _bnb_warm_worker(rank, world, port, out):
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    X, y, P = _bnb_problem()
    fake = _FakeSnapCtx(X, y, P)
    res = pls.dist.bnb_search_warm(fake, P.shape[1] + 1, rank=rank, world=world, batch=4)
    out[rank] = (res, len(fake.live), fake.handed, fake.warm, fake.cold)
    dist.destroy_process_group()


def test_bnb_search_warm_protocol_single_and_gloo_world2(partls, oracle):
    """bnb_search_warm (nodes dealt to the rank that holds their parent's snapshot): same optimum as the cold search and the oracle;
    the snapshot protocol is sound — no slot used after release, none released twice, none leaked at the end of a complete search —
    with one rank, with a pool too small for the frontier, and with 2 gloo ranks (where both ranks end up owning subtrees)."""
    import torch.multiprocessing as mp
    X, y, P = _bnb_problem()
    ref = oracle.fit_bnb(X, y, P)
    cold = partls.dist.bnb_search(_oracle_bound_fn(X, y, P), P.shape[1] + 1, batch=4)
    for cap in (10_000, 3):
        fake = _FakeSnapCtx(X, y, P, capacity=cap)
        mu, pat, free, bounded = partls.dist.bnb_search_warm(fake, P.shape[1] + 1, batch=4)
        assert abs(mu - ref["opt"]) <= 1e-9 * max(1.0, ref["opt"]) and (pat, free) == (cold[1], cold[2])
        assert len(fake.live) == 0 and fake.handed > 0 and fake.warm > 0          # every snapshot was returned; warm starts happened
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_bnb_warm_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(2):
        (mu, pat, free, bounded), live, handed, warm, cold_n = out[r]
        assert (pat, free) == (cold[1], cold[2]) and abs(mu - cold[0]) <= 1e-12 * max(1.0, cold[0])
        assert live == 0 and warm > 0                                             # both ranks own snapshots and use them
    assert out[0][0][3] == out[1][0][3]


class _FakeCtx:
    """stands in for a Context in dist.reduce_winner: hands out a shard's candidates, records the merged list it is given and merges it
    the way partls_opt_merge_candidates does (lexicographic minimum; near ties within `window` of the winner's objective^2)"""

    def __init__(self, cands, window):
        self.cands, self.window, self.merged = cands, window, None

    def opt_candidates(self):
        import numpy as np
        return np.array([c[0] for c in self.cands]), np.array([c[1] for c in self.cands], dtype=np.int64)

    def opt_merge_candidates(self, objs, pats):
        allc = sorted({(float(o), int(p)) for o, p in zip(objs, pats) if p >= 0})
        self.merged = allc
        if not allc:
            return float("inf"), -1
        win = allc[0]
        self.near = [p for o, p in allc[1:] if o * o <= win[0] ** 2 + self.window][:3]
        return win


def _winner_worker(rank, world, port, cases, out):
    import torch.distributed as dist
    import partls_amd
    pls = partls_amd.package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = []
    for per_rank, window in cases:
        ctx = _FakeCtx(per_rank[rank], window)
        obj, pat = (per_rank[rank][0] if per_rank[rank] else (float("inf"), -1))
        w = pls.dist.reduce_winner(ctx, obj, pat, order_key=3.0)
        res.append((w, ctx.merged, getattr(ctx, "near", None)))
    out[rank] = res
    dist.destroy_process_group()


def test_reduce_winner_gathers_every_shards_near_ties_gloo_world2(partls):
    """dist.reduce_winner: one all-reduce(min) on the objective + one all-gather of every rank's winner and near ties.  Both ranks must end
    with the SAME merged candidate list — the winner may sit on one rank and its near tie on the other — and hence install the same set."""
    import torch.multiprocessing as mp
    cases = [
        ([[(2.0, 9), (2.0000001, 4)], [(2.00000005, 70)]], 1e-6),          # winner on rank 0, near ties on both ranks
        ([[(3.0, 1)], [(1.0, 5), (1.0, 6)]], 1e-9),                        # winner and an exact tie on rank 1; rank 0 far away
        ([[], [(4.0, 2)]], 1e-9),                                          # rank 0's shard was empty
        ([[(1.5, 1 << 40)], [(1.5, (1 << 40) - 1)]], 0.0),                 # exact tie across ranks: first index; indices beyond 2^32
    ]
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_winner_worker, args=(r, 2, port, cases, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = out[0], out[1]
    assert [x[0] for x in r0] == [x[0] for x in r1] == [(2.0, 9), (1.0, 5), (4.0, 2), (1.5, (1 << 40) - 1)]
    for a, b in zip(r0, r1):
        assert a[1] == b[1] and a[2] == b[2]
    assert r0[0][1] == [(2.0, 9), (2.00000005, 70), (2.0000001, 4)] and r0[0][2] == [70, 4]
    assert r0[1][2] == [6] and r0[3][2] == [1 << 40]
