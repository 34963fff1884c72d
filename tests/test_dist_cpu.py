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


def test_ray_bounds_host_logic():
    """ray_tracing.sphere_entry_exit (torch glue of RayTracing, no kernel) == the oracle's restatement of
    intersection_with_unit_sphere, hits and tangent-plane misses, cameras outside and inside."""
    import torch
    from oracle import iso_oracle as O
    from iso_points_amd.ray_tracing import sphere_entry_exit
    g = torch.Generator().manual_seed(0)
    for cam in ([[0.0, 0.3, 2.5]], [[0.2, -0.1, 0.4]], [[0.0, 0.0, -3.0], [1.5, 1.5, 0.0]]):
        c = torch.tensor(cam)
        d = torch.nn.functional.normalize((torch.rand(c.shape[0], 500, 3, generator=g) - 0.5) * 2.5 - c[:, None], dim=-1)
        for radius in (1.0, 0.7):
            e0, e1, hit = sphere_entry_exit(c, d, radius)
            r0, r1, rh = O.sphere_entry_exit(c, d, radius)
            assert torch.equal(hit, rh) and 0 < int(hit.sum()) < hit.numel() or c.norm(dim=-1).min() < radius
            assert torch.allclose(e0, r0, rtol=0, atol=2e-6) and torch.allclose(e1, r1, rtol=0, atol=2e-6)
