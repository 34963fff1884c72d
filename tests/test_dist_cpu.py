"""Multi-process (gloo, world_size 2/3, CPU) tests of the sharding helpers the N>1 path is
built from (iso_points_amd/dist.py): shard bounds, ragged all-gather in shard order, reductions."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iso_points_amd.dist import Comm, shard_bounds
        c = Comm()
        assert c.world == world and c.rank == rank
        full = torch.arange(n_total * 3, dtype=torch.float32).view(n_total, 3)
        lo, hi = shard_bounds(n_total, world, rank)
        got = c.all_gather_rows(full[lo:hi].clone(), n_total)
        ok = torch.equal(got, full)
        # sum-reduce of disjoint segments == concatenation (how h / occ_grad bands are merged)
        buf = torch.zeros(n_total)
        buf[lo:hi] = full[lo:hi, 0]
        c.all_reduce_(buf, "sum")
        ok = ok and torch.equal(buf, full[:, 0])
        flags = torch.zeros(n_total, dtype=torch.int32)
        flags[rank::world] = 1
        c.all_reduce_(flags, "max")
        ok = ok and bool((flags == 1).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (3, 8)])
def test_comm_helpers_gloo(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert all(res.values())


def test_shard_bounds_cover_range():
    from iso_points_amd.dist import all_shard_bounds, shard_bounds
    for n in (0, 1, 7, 32, 1000001):
        for w in (1, 2, 3, 8):
            b = all_shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
            assert b[w - 1] == shard_bounds(n, w, w - 1)
