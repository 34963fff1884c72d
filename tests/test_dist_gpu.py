"""The sharded iso-point cycle (iso_points_amd/dist.py) with 2 ranks == the single-GPU cycle.
Both ranks share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one
device); the collectives, shard bookkeeping and every kernel are the ones the N>1 bench runs."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(dev, P, S, n_views=3):
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.dist import sphere_silhouette
    from iso_points_amd.rasterizer import PointsRasterizationSettings
    from iso_points_amd.sdf_models import Siren
    g = torch.Generator().manual_seed(0)
    pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    pts = (pts + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)
    views = torch.stack([look_at_view(5.0, 20.0, 120.0 * i) for i in range(n_views)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    rs = PointsRasterizationSettings(image_size=S, points_per_pixel=6)
    target = sphere_silhouette(S, n_views, 5.0, 30.0, dev)
    return pts, views, projs, rs, target


def _run(model_kind, dev, comm, P, S, n_views=3):
    from iso_points_amd.dist import IsoCycle
    from iso_points_amd.sdf_models import SphereSDF, Siren
    pts, views, projs, rs, target = _setup(dev, P, S, n_views)
    if model_kind == "sphere":
        model = SphereSDF().to(dev)
    else:
        torch.manual_seed(0)
        model = Siren(hidden_size=128, n_layers=2).to(dev)     # random weights: fixed iteration counts
    cyc = IsoCycle(model, pts, views, projs, raster_settings=rs, comm=comm, target=target)
    return cyc, cyc.step()


def _worker(rank, world, port, model_kind, P, S, outdir, n_views=3):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iso_points_amd.dist import Comm
        dev = torch.device("cuda:0")
        comm = Comm()
        cyc, (r1, img, grad, frags, filt) = _run(model_kind, dev, comm, P, S, n_views)
        pts_all = comm.all_gather_rows(r1.points[0], P)
        # merge the bands / slices so that every rank holds the full result
        idx = frags.idx.clone(); comm.all_reduce_(idx, "max")           # -1 outside the own band
        zb = frags.zbuf.clone(); comm.all_reduce_(zb, "max")
        im = img.clone(); comm.all_reduce_(im, "sum")                    # 0 outside the own band
        gxy = grad[:, :2].clone().contiguous(); comm.all_reduce_(gxy, "sum")   # 0 outside the own slices
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"pts": pts_all.cpu(), "idx": idx.cpu(), "zbuf": zb.cpu(), "img": im.cpu(),
                        "gxy": gxy.cpu(), "gz": grad[:, 2].cpu(), "knn": None}, os.path.join(outdir, "sharded.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model_kind,n_views,world", [("sphere", 3, 2), ("siren", 3, 2), ("sphere", 2, 2),
                                                      ("sphere", 2, 4)])
def test_sharded_cycle_equals_single_gpu(dev, tmp_path, model_kind, n_views, world):
    """3 views on 2 ranks: every rank queries a row range of every view; 2 views on 2 ranks: a view
    belongs to one rank; 2 views on 4 ranks: to two ranks (what the 8-GPU x 4-view bench runs)."""
    from iso_points_amd.dist import Comm
    P, S = 30001, 80            # odd P: uneven shards; S not a multiple of 16*world
    cyc, (r1, img, grad, frags, filt) = _run(model_kind, dev, Comm(enabled=False), P, S, n_views)
    ref = {"pts": r1.points[0].cpu(), "idx": frags.idx.cpu(), "zbuf": frags.zbuf.cpu(), "img": img.cpu(),
           "gxy": grad[:, :2].cpu(), "gz": grad[:, 2].cpu()}
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model_kind, P, S, str(tmp_path), n_views))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = torch.load(os.path.join(str(tmp_path), "sharded.pt"))
    assert torch.equal(got["pts"], ref["pts"])          # projection + FRNN + repulsion: bit-identical
    assert torch.equal(got["idx"], ref["idx"])          # per-pixel splat lists: bit-identical
    assert torch.equal(got["zbuf"], ref["zbuf"])
    assert torch.equal(got["img"], ref["img"])
    assert torch.equal(got["gxy"], ref["gxy"])          # point-major xy gradient: bit-identical
    # z gradient: pixel-major atomic scatter on the sharded path vs point-major sum on one GPU
    scale = ref["gz"].abs().max().clamp_min(1e-30)
    assert ((got["gz"] - ref["gz"]).abs().max() / scale) < 1e-5
    assert ref["gz"].abs().sum() > 0 and ref["gxy"].abs().sum() > 0


class _RankOf(object):
    """stand-in for Comm: rank `rank` of `world`, no process group (only .world / .rank are read)"""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank


@pytest.mark.parametrize("world,n_views", [(4, 2), (8, 4), (4, 4), (3, 4), (6, 4), (2, 1)])
def test_h_shares_sum_to_the_whole(dev, world, n_views):
    """IsoCycle._h_share over all ranks (what the sum all-reduce adds up) == the all-rows result,
    for view-owned (world a multiple of the views) and row-range (otherwise) splits."""
    from iso_points_amd.dist import IsoCycle
    from iso_points_amd.levelset_sampling import with_host_lengths
    from iso_points_amd.sdf_models import SphereSDF
    pts, views, projs, rs, target = _setup(dev, 20011, 64, n_views)
    cyc = IsoCycle(SphereSDF().to(dev), pts, views, projs, raster_settings=rs, target=target)
    p_all = torch.nn.functional.normalize(pts[0], dim=-1)
    flags, off, lens = cyc.splat.filter_renderable(p_all, p_all, cyc.views)
    tot = sum(lens)
    fl = [sum(lens[:i]) for i in range(n_views)]
    num = with_host_lengths(torch.tensor(lens, dtype=torch.int64, device=dev), lens)
    pts_f = cyc.splat.compact(p_all, flags, off, p_all.shape[0], tot)
    cyc.comm = _RankOf(n_views + 1, 0) if n_views > 1 else _RankOf(3, 0)     # a row-range split ...
    ref = torch.zeros(tot, device=dev)
    for r in range(cyc.comm.world):                                          # ... summed = the whole
        cyc.comm.rank = r
        ref += cyc._h_share(pts_f, lens, fl, num, tot)
    assert (ref > 0).all()
    total = torch.zeros(tot, device=dev)
    filled = torch.zeros(tot, device=dev)
    for r in range(world):
        cyc.comm = _RankOf(world, r)
        h = cyc._h_share(pts_f, lens, fl, num, tot)
        total += h
        filled += (h != 0).float()
    assert torch.equal(total, ref) and (filled == 1).all()                   # every row written by one rank
