"""The sharded iso-point cycle (iso_points_amd/dist.py) against the single-GPU one.

world ranks in lock-step inside one process (run_lockstep: the same generator code the process group
drives): every result of the N-rank job must be BIT-identical to the single-GPU cycle on the same
(x-slab ordered) cloud -- projected points, per-pixel index lists, z-buffers, images and the gradients
of every packed row (z included: fixed-point accumulation).  Plus one real 2-process run over gloo."""
import os
import subprocess
import sys

import pytest
import torch

from util import sphere_cloud

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(dev, P, S, N, seed=3):
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.dist import sphere_silhouette
    from iso_points_amd.rasterizer import PointsRasterizationSettings
    pts = sphere_cloud(P, seed=seed).to(dev)
    views = torch.stack([look_at_view(3.0, 20.0, 360.0 / N * i) for i in range(N)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    rs = PointsRasterizationSettings(image_size=S, points_per_pixel=8, cutoff_threshold=1.0, depth_merging_threshold=0.05,
                                     radii_backward_scaler=10, backface_culling=True, Vrk_isotropic=True)
    return pts, views, projs, rs, sphere_silhouette(S, N, 3.0, 30.0, dev)


def _models(dev, kind):
    from iso_points_amd.sdf_models import SphereSDF
    if kind == "sphere":
        return SphereSDF().to(dev)
    from oracle import iso_oracle as O
    from util import fitted_siren
    return fitted_siren(O, 128, 2, seed=0, fit=100).to(dev)


@pytest.mark.parametrize("world,P,S,N,kind", [(2, 20000, 64, 2, "sphere"), (4, 60000, 128, 4, "sphere"),
                                              (8, 200000, 256, 4, "sphere"), (3, 30000, 96, 3, "siren"),
                                              (8, 9000, 64, 1, "sphere"), (3, 20011, 64, 2, "sphere"),
                                              (5, 50021, 80, 3, "sphere"), (7, 40000, 112, 5, "sphere"),
                                              (6, 24000, 48, 6, "siren")])
def test_lockstep_ranks_equal_single_gpu(dev, world, P, S, N, kind):
    from iso_points_amd.dist import IsoCycle, run_lockstep, shard_bounds, slab_order
    pts, views, projs, rs, target = _scene(dev, P, S, N)
    model = _models(dev, kind)
    pts = pts[:, slab_order(pts[0], world)].contiguous()
    one = IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target)
    r1, img, grad, frags, fr = one.step()
    one.check(fr)
    tot = int(fr["num_points"].sum().item())
    ranks = [IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target, world=world, rank=r)
             for r in range(world)]
    res = run_lockstep(ranks)
    for r, (c, out) in enumerate(zip(ranks, res)):
        u = c.check(out[4])
        assert u["halo_exported"] > 0 and u["grid"]["overflow_bricks"] == 0
    for r, (c, (q1, qimg, qgrad, qfrags, qfr)) in enumerate(zip(ranks, res)):
        lo, hi = shard_bounds(P, world, r)
        assert torch.equal(q1.points[0], r1.points[0, lo:hi]), "rank %d: resampled points differ" % r
        assert torch.equal(q1.normals[0], r1.normals[0, lo:hi]) and torch.equal(q1.mask[0], r1.mask[0, lo:hi])
        # the global packed layout every rank derives from the all-gathered counts = the single-GPU one
        assert torch.equal(qfr["first_global"], fr["first_idx"]) and torch.equal(qfr["num_global"], fr["num_points"])
        seen = ((fr["mask"][None, lo:hi] >> torch.arange(N, device=dev)[:, None]) & 1).bool()      # h exists where the point is rendered
        assert torch.equal(qfr["mask"], fr["mask"][lo:hi]) and torch.equal(qfr["h"][seen], fr["h"][:, lo:hi][seen]), \
            "rank %d: bandwidths differ" % r
        # the rank's own packed rows = its slice of every view of the single-GPU arrays
        for v in range(N):
            a, n, lf = int(qfr["own_first"][v]), int(qfr["own_num"][v]), int(qfr["local_first"][v])
            for k in ("ndc", "ellipse_params", "radii", "scaler", "features"):
                assert torch.equal(qfr["own"][k][lf:lf + n], fr[k][a:a + n]), "rank %d view %d: packed %s differs" % (r, v, k)
        # the rows the band received: ascending global ids inside every view, each one the single-GPU row of that id
        nl = [int(x) for x in qfr["num_points"].tolist()]
        fl = [int(x) for x in qfr["first_idx"].tolist()]
        for v in range(N):
            g = c.gid[fl[v]:fl[v] + nl[v]].long()
            assert (g[1:] > g[:-1]).all() and (nl[v] == 0 or (g[0] >= int(fr["first_idx"][v]) and
                                                               g[-1] < int(fr["first_idx"][v] + fr["num_points"][v])))
            for k in ("ndc", "ellipse_params", "radii", "scaler", "features"):
                assert torch.equal(qfr[k][fl[v]:fl[v] + nl[v]], fr[k][g]), "rank %d view %d: received %s differs" % (r, v, k)
        y0, y1 = c.band_rows()
        if y1 > y0:
            assert torch.equal(qfrags.idx[:, y0:y1], frags.idx[:, y0:y1]), "rank %d: index lists differ" % r
            assert torch.equal(qfrags.zbuf[:, y0:y1], frags.zbuf[:, y0:y1])
            assert torch.equal(qfrags.qvalue[:, y0:y1], frags.qvalue[:, y0:y1])
            assert torch.equal(qfrags.occupancy[:, y0:y1], frags.occupancy[:, y0:y1])
            assert torch.equal(qimg[:, y0:y1], img[:, y0:y1]), "rank %d: image band differs" % r
        for v in range(N):
            a, n, lf = int(qfr["own_first"][v]), int(qfr["own_num"][v]), int(qfr["local_first"][v])
            assert torch.equal(qgrad[lf:lf + n], grad[a:a + n]), "rank %d view %d: row gradients differ" % (r, v)
    # the own rows of all ranks tile the packed layout
    cover = torch.zeros(tot, dtype=torch.int32)
    for (_, _, _, _, qfr) in res:
        for v in range(N):
            a, n = int(qfr["own_first"][v]), int(qfr["own_num"][v])
            cover[a:a + n] += 1
    assert (cover == 1).all()


@pytest.mark.parametrize("world", [1, 4])
def test_graph_replay_equals_eager(dev, world):
    """use_graphs: the segments between two exchanges replayed as HIP graphs give the eager results,
    step after step (nothing in the cycle depends on a host read)."""
    from iso_points_amd.dist import IsoCycle, run_lockstep, slab_order
    P, S, N = 50000, 128, 2
    pts, views, projs, rs, target = _scene(dev, P, S, N)
    model = _models(dev, "sphere")
    pts = pts[:, slab_order(pts[0], world)].contiguous()
    ranks = [IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target, world=world, rank=r)
             for r in range(world)]
    eager = run_lockstep(ranks)                                                                # also the warm-up
    from iso_points_amd.rasterizer import PointFragments
    eager = [(a[0], a[1].clone(), a[2].clone(),                      # (N ranks: the arrays are the cycle's persistent buffers)
              PointFragments(a[3].idx.clone(), a[3].zbuf.clone(), a[3].qvalue, None, a[3].occupancy)) for a in eager]
    for c in ranks:
        c.use_graphs = True
        c.marks = True
    run_lockstep(ranks)                                                                        # capture
    for _ in range(3):
        got = run_lockstep(ranks)                                                              # replays
        for a, b in zip(eager, got):
            assert torch.equal(a[0].points, b[0].points) and torch.equal(a[1], b[1])
            # the row gradients over the rows that exist (the packed arrays have capacity size; rows beyond the
            # device-side totals are unspecified)
            for f0, n0 in zip(b[4].get("local_first", b[4]["own_first"]).tolist(), b[4]["own_num"].tolist()):
                assert torch.equal(a[2][f0:f0 + n0], b[2][f0:f0 + n0])
            assert torch.equal(a[3].idx, b[3].idx) and torch.equal(a[3].zbuf, b[3].zbuf)
    for c, o in zip(ranks, got):
        c.check(o[4])


def test_calibrated_capacities(dev):
    """calibrate() shrinks the exchange buffers; the cycle afterwards is unchanged and check() passes."""
    from iso_points_amd.dist import IsoCycle, run_lockstep
    from iso_points_amd.dist import slab_order
    world, P = 4, 40000
    pts, views, projs, rs, target = _scene(dev, P, 64, 2)
    model = _models(dev, "sphere")
    pts = pts[:, slab_order(pts[0], world)].contiguous()
    ranks = [IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target, world=world, rank=r)
             for r in range(world)]
    base = run_lockstep(ranks)
    use = [c.check(o[4]) for c, o in zip(ranks, base)]
    for c in ranks:      # what calibrate() does, without a process group
        c.halo_cap = max(1024, int(1.5 * max(u["halo_exported"] for u in use)))
        c.import_cap = max(1024, int(1.5 * max(u["halo_imported"] for u in use)))
        c.rec_cap = max(1024, int(1.25 * max(u["own_rows"] for u in use)))
        c.seg_cap = max(1024, int(1.25 * max(u["segment_records"] for u in use)))
        c._alloc()
    again = run_lockstep(ranks)
    for c, a, b in zip(ranks, base, again):
        u = c.check(b[4])
        assert u["segment_records"] <= c.seg_cap and u["band_rows"] <= c.cap_local
        assert torch.equal(a[0].points, b[0].points) and torch.equal(a[3].idx, b[3].idx)
        for f0, n0 in zip(b[4]["local_first"].tolist(), b[4]["own_num"].tolist()):      # the rows that exist
            assert torch.equal(a[2][f0:f0 + n0], b[2][f0:f0 + n0])
    # a capacity that is too small is reported, not silently dropped
    for c in ranks:
        c.halo_cap = 16
        c._alloc()
    out = run_lockstep(ranks)
    with pytest.raises(RuntimeError):
        for c, o in zip(ranks, out):
            c.check(o[4])
    # ... and so is a band segment that is too small
    for c in ranks:
        c.halo_cap = max(1024, int(1.5 * max(u["halo_exported"] for u in use)))
        c.seg_cap = 64
        c._alloc()
    out = run_lockstep(ranks)
    with pytest.raises(RuntimeError, match="band_overflow"):
        for c, o in zip(ranks, out):
            c.check(o[4])


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
dist.init_process_group(backend="gloo")
from test_dist_gpu import _scene, _models
from iso_points_amd.dist import Comm, IsoCycle, slab_order
dev = torch.device("cuda:0")
world, rank = dist.get_world_size(), dist.get_rank()
pts, views, projs, rs, target = _scene(dev, 30000, 96, 2)
model = _models(dev, "sphere")
pts = pts[:, slab_order(pts[0], world)].contiguous()
c = IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target, comm=Comm())
c.calibrate()
r1, img, grad, frags, fr = c.step()
c.check(fr)
one = IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target)
s1, simg, sgrad, sfrags, sfr = one.step()
y0, y1 = c.band_rows()
ok = torch.equal(r1.points[0], s1.points[0, c.lo:c.hi]) and torch.equal(frags.idx[:, y0:y1], sfrags.idx[:, y0:y1]) \
    and torch.equal(img[:, y0:y1], simg[:, y0:y1])
for v in range(2):
    a, n, lf = int(fr["own_first"][v]), int(fr["own_num"][v]), int(fr["local_first"][v])
    ok = ok and torch.equal(grad[lf:lf + n], sgrad[a:a + n])
print("RANK", rank, "OK" if ok else "MISMATCH", c.comm.bytes_log)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_two_processes_over_gloo(dev, tmp_path):
    """The same cycle under a real process group (2 processes sharing the test GPU, gloo)."""
    w = tmp_path / "worker.py"
    w.write_text(_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(w), ROOT],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("OK") == 2


@pytest.mark.parametrize("launcher", [True, False], ids=["torchrun", "bare"])
def test_bench_line_of_two_ranks(dev, launcher):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per process;
    both ranks on the test GPU and gloo instead of RCCL through the bench's own test hook): the cycle runs with
    calibrated buffers, graph replay and the timing protocol, and rank 0 prints the contract's line."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ISO_BENCH_ONE_DEVICE="1", ISO_BENCH_BACKEND="gloo")
    cmd = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if launcher:
        cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29533"] + cmd
    else:
        env.pop("WORLD_SIZE", None)           # bare `python bench.py --gpus 2`: bench.py starts its own ranks
    out = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["parallelism"] == "2 ranks" or "2" in d["config"]["parallelism"]


def test_standalone_project_resample_leaves_no_pending_box(dev):
    """project_resample() driven on its own (bench.active_counts, tools) on a follow-capable model used to
    leave the second projection's bounding box pending in the grid's workspace; the next Follow projection then unioned
    its box with the stale one and the grid's radius / spacing silently differed from the no-follow route.  Two
    stand-alone calls, then a whole cycle, must equal the no_follow route bit for bit -- and a box left behind by an
    aborted sequence is dropped."""
    from iso_points_amd import bricks
    from iso_points_amd.dist import IsoCycle, slab_order, _Single
    pts, views, projs, rs, tgt = _scene(dev, 30000, 128, 2)
    pts = pts[:, slab_order(pts[0], 1)].contiguous()
    model = _models(dev, "sphere")
    a = IsoCycle(model, pts, views, projs, raster_settings=rs, comm=_Single(), target=tgt)
    b = IsoCycle(model, pts, views, projs, raster_settings=rs, comm=_Single(), target=tgt)
    b.no_follow = True
    for _ in range(2):
        ra, rb = a.run(a.project_resample()), b.run(b.project_resample())
        assert torch.equal(ra.points, rb.points) and torch.equal(ra.normals, rb.normals)
        assert not a._box_pending
    # an aborted sequence: a Follow projection whose build never ran
    from iso_points_amd.sdf_models import SphereSDF
    f = bricks.Follow(a.grid, a.n_own)
    small, a.model = a.model, SphereSDF(radius=3.0).to(dev)
    a.run(a._project(a.pts0_local, 10, follow=f))                # the box of a sphere three times as large, left pending
    a.model = small
    assert f.done
    a._box_pending = True
    ca, cb = a.run(a.cycle()), b.run(b.cycle())
    assert torch.equal(ca[0].points, cb[0].points), "resampled points differ"
    assert torch.equal(ca[1], cb[1]), "images differ"
    tot = int(ca[4]["num_points"].sum().item())
    assert torch.equal(ca[2][:tot], cb[2][:tot]), "gradients differ"
    ha, hb = a.grid.header(), b.grid.header()
    assert ha == hb, (ha, hb)
