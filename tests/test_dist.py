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
