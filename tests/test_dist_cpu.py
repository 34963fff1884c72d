"""Multi-process (gloo, world_size 2/3, CPU) tests of what the N>1 path is built from
(iso_points_amd/dist.py): shard bounds, the x-slab order, the exchange protocol of `Comm.execute`
(the requests the cycle generator yields) and the in-process lock-step driver, which must give the
same answers as the process group."""
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


def _toy_cycle(rank, world, n):
    """A generator with the exchange pattern of IsoCycle.cycle: gather of boxes, gather of a padded
    record buffer with a count in word 0, max / sum reductions of disjoint segments."""
    box = torch.tensor([float(rank), 0, 0, 0, float(rank + 1), 0, 0, 0])
    boxes = yield ("all_gather", box)
    cap = 5
    buf = torch.zeros(cap + 1)
    cnt = 1 + rank % cap
    buf[0] = cnt
    buf[1:1 + cnt] = torch.arange(cnt, dtype=torch.float32) + 10 * rank
    got = yield ("all_gather", buf)
    vis = torch.zeros(n, dtype=torch.uint8)
    vis[rank::world] = 1
    vis = yield ("all_reduce", vis, "max")
    acc = torch.zeros(n, dtype=torch.int64)
    acc[rank::world] = rank + 1
    acc = yield ("all_reduce", acc, "sum")
    return boxes, got, vis, acc


def _expect(world, n):
    boxes = torch.stack([torch.tensor([float(r), 0, 0, 0, float(r + 1), 0, 0, 0]) for r in range(world)])
    vis = torch.ones(n, dtype=torch.uint8)
    acc = torch.tensor([(i % world) + 1 for i in range(n)], dtype=torch.int64)
    return boxes, vis, acc


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iso_points_amd.dist import Comm
        c = Comm()
        assert c.world == world and c.rank == rank
        g = _toy_cycle(rank, world, n_total)
        try:
            req = next(g)
            while True:
                req = g.send(c.execute(req))
        except StopIteration as e:
            boxes, got, vis, acc = e.value
        eb, ev, ea = _expect(world, n_total)
        ok = torch.equal(boxes, eb) and torch.equal(vis, ev) and torch.equal(acc, ea)
        for r in range(world):
            cnt = int(got[r, 0])
            ok = ok and cnt == 1 + r % 5 and torch.equal(got[r, 1:1 + cnt], torch.arange(cnt, dtype=torch.float32) + 10 * r)
        ok = ok and [k for k, _ in c.bytes_log] == ["all_gather", "all_gather", "all_reduce", "all_reduce"]
        ok = ok and c.max_int(rank * 7, torch.device("cpu")) == (world - 1) * 7
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (3, 8)])
def test_exchange_protocol_gloo(world, n_total):
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


@pytest.mark.parametrize("world,n_total", [(2, 10), (3, 8), (8, 33)])
def test_lockstep_driver_matches_the_protocol(world, n_total):
    """run_lockstep's hand-made exchanges = what the process group returns (same toy cycle)."""
    from iso_points_amd.dist import run_lockstep

    class Toy(object):
        def __init__(self, r):
            self.r = r

        def cycle(self):
            return _toy_cycle(self.r, world, n_total)

    res = run_lockstep([Toy(r) for r in range(world)])
    eb, ev, ea = _expect(world, n_total)
    for r, (boxes, got, vis, acc) in enumerate(res):
        assert torch.equal(boxes, eb) and torch.equal(vis, ev) and torch.equal(acc, ea)
        for s in range(world):
            assert int(got[s, 0]) == 1 + s % 5


def test_shard_bounds_cover_everything():
    from iso_points_amd.dist import all_shard_bounds, shard_bounds
    for n in (0, 1, 7, 100, 1000003):
        for w in (1, 2, 3, 8):
            b = all_shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
            assert shard_bounds(n, w, w - 1) == b[-1]


def test_slab_order_is_a_stable_x_sort():
    from iso_points_amd.dist import shard_bounds, slab_order
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1000, 3, generator=g)
    p[100:120, 0] = p[5, 0]                      # ties keep their original order
    assert torch.equal(slab_order(p, 1, local=None), torch.arange(1000))
    perm = slab_order(p, 4, local=None)
    xs = p[perm, 0]
    assert (xs[1:] >= xs[:-1]).all() and sorted(perm.tolist()) == list(range(1000))
    tie = perm[(xs == p[5, 0])]
    assert (tie[1:] > tie[:-1]).all()
    # rank r's slab lies left of rank r+1's
    for r in range(3):
        lo, hi = shard_bounds(1000, 4, r)
        assert xs[hi - 1] <= xs[hi]


def test_slab_order_cell_order_keeps_the_slabs_and_is_local():
    """Default order: the same x-slabs (same point sets per rank as the plain x sort), rows of a slab along the
    z-order curve -- consecutive rows are near each other."""
    from iso_points_amd.dist import shard_bounds, slab_order
    g = torch.Generator().manual_seed(1)
    p = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    for world in (1, 3, 4):
        by_x = slab_order(p, world, local=None)
        perm = slab_order(p, world)
        assert sorted(perm.tolist()) == list(range(20000))
        for r in range(world):
            lo, hi = shard_bounds(20000, world, r)
            assert set(perm[lo:hi].tolist()) == set(by_x[lo:hi].tolist())
        q = p[perm]
        step = (q[1:] - q[:-1]).norm(dim=-1).median()
        assert step < 0.1 * (p[1:] - p[:-1]).norm(dim=-1).median()
    assert torch.equal(slab_order(p, 2), slab_order(p.clone(), 2))        # deterministic


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started WITHOUT torch.distributed.run (the driver's recorded command form for one
    GPU, extended to N): bench.py re-executes itself under the launcher, one process per rank, and rank 0 prints one
    line.  ISO_BENCH_DRYRUN keeps the ranks to rendezvous + the timing protocol's collectives (gloo; no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISO_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["steps"] == 3 and d["max_over_ranks"] == 2.0
