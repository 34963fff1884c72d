"""BASELINE.json configs[3] and configs[4] at their own sizes (the oracle cannot run them in seconds:
size-independent properties), plus the sharded path with the IDR network and the insert / splat chain
of configs[4] against the oracle at reduced size.

  configs[3]: 4 M points, 8-layer IDR SDF (8 x 512, skip 4, 6 frequencies), points sharded by brick slab
  configs[4]: 500 k iso-points, loss-weighted insert, splat fwd+bwd at the reference's largest squares
              (1024, 1344: rasterizer.py:52 is square-only and rasterize_points.cu:462 caps the bins) and at
              1200 x 1600 (H != W: pytorch3d's non-square NDC convention, tests/test_splat_gpu.py pins it to
              the square sub-case)"""
import pytest
import torch

from util import sphere_cloud

pytestmark = pytest.mark.gpu


def _idr(dev, scale=1.0):
    from oracle import iso_oracle as O
    torch.manual_seed(4)
    return O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6).to(dev)


def test_cfg3_idr_4m_projection_properties(dev):
    """4 M points through the fused IDR Newton projection: every point flagged converged satisfies
    |sdf| <= tol under an INDEPENDENT evaluation (fused value+gradient kernel on the returned points), its
    normal is the gradient there, and a converged cloud does not move when projected again."""
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import idr_sdf_and_grad
    P, tol = 4000000, 5e-5
    m = _idr(dev)
    g = torch.Generator().manual_seed(40)
    pts = (0.6 * torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
           + 0.04 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)      # geometric init = sphere of radius ~0.6
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=tol, knn_k=8)
    r = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    mask = r.mask[0].bool()
    assert float(mask.float().mean()) > 0.95
    chunk = 1000000
    for a in range(0, P, chunk):
        sl = slice(a, a + chunk)
        sdf, grad = idr_sdf_and_grad(m, r.points[0, sl])
        mk = mask[sl]
        assert float(sdf[mk].abs().max()) <= tol
        assert torch.equal(grad[mk], r.normals[0, sl][mk])
    r2 = proj._project_points(m, r.points, full_lengths(pts), proj_max_iters=10)
    assert torch.equal(r2.points[0][mask], r.points[0][mask])


@pytest.mark.parametrize("world", [2, 8])
def test_cfg3_sharded_idr_project_resample_is_bit_identical(dev, world):
    """The point stages of the cycle (project T=10, halo exchange, fused FRNN + repulsion, project T=3) with
    the 8 x 512 IDR network on `world` slab shards = the single-GPU result, bit for bit."""
    from iso_points_amd.dist import IsoCycle, run_lockstep, shard_bounds, slab_order
    from iso_points_amd.cameras import look_at_view, perspective
    P = 160000
    m = _idr(dev)
    g = torch.Generator().manual_seed(41)
    pts = (0.6 * torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
           + 0.03 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)
    pts = pts[:, slab_order(pts[0], world)].contiguous()
    views = torch.stack([look_at_view(3.0, 20.0, 0.0)]).to(dev)
    projs = views @ perspective(30.0).to(dev)

    class PR(object):                      # only stages 1 and 2 of the cycle
        def __init__(self, c):
            self.c = c

        def cycle(self):
            return self.c.project_resample()
    one = IsoCycle(m, pts, views, projs, knn_k=8)
    ref = one.run(one.project_resample())
    ranks = [IsoCycle(m, pts, views, projs, knn_k=8, world=world, rank=r) for r in range(world)]
    res = run_lockstep([PR(c) for c in ranks])
    for r, (c, q) in enumerate(zip(ranks, res)):
        lo, hi = shard_bounds(P, world, r)
        assert torch.equal(q.points[0], ref.points[0, lo:hi]) and torch.equal(q.normals[0], ref.normals[0, lo:hi])
        assert torch.equal(q.mask[0], ref.mask[0, lo:hi])
        u = c.usage()
        assert u["halo_uncertified"] == 0 and u["halo_export_overflow"] == 0 and u["halo_import_overflow"] == 0


def test_cfg4_insert_500k_and_large_square_splat(dev):
    """500 k iso-points: loss-weighted insert around 5 000 FPS reference points (levelset_sampling.py:172-233),
    then splat forward + compositing + backward at 1024^2 and 1344^2: structural properties of the results."""
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.point_processing import farthest_sampling
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting, _C, _visible_and_radius, composite
    P = 500000
    pts = torch.nn.functional.normalize(sphere_cloud(P, seed=50), dim=-1).to(dev)
    num = full_lengths(pts)
    ref = farthest_sampling(pts, num, 5000 / P)[0][0].contiguous()
    assert ref.shape == (5000, 3)
    g = torch.Generator().manual_seed(51)
    metric = torch.exp(3 * torch.randn(5000, 1, generator=g)).to(dev)

    class Ref(object):
        def points_packed(self): return ref
        def features_packed(self): return metric
        def num_points_per_cloud(self): return torch.tensor([5000], device=dev)
    proj = UniformProjection(knn_k=8)
    new_pts, new_num, child, child_n = proj.insert(Ref(), pts, num)
    n_child = int(child_n.item())
    assert 0 < n_child <= 8 * P and new_pts.shape[1] == P + n_child
    # children come in groups of 8 around a father (2/3 father + 1/3 neighbour): a group is tighter than the
    # neighbour search radius
    grp = child[0].view(-1, 8, 3)
    assert float((grp - grp.mean(dim=1, keepdim=True)).norm(dim=-1).max()) < 0.2
    cloud = torch.nn.functional.normalize(new_pts[0], dim=-1).contiguous()
    views = torch.stack([look_at_view(3.0, 20.0, 0.0)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    for S in (1024, 1344, (1200, 1600)):      # the last one is configs[4]'s own frame: beyond the reference (H != W)
        rs = PointsRasterizationSettings(image_size=S, points_per_pixel=8)
        ss = SurfaceSplatting(raster_settings=rs)
        frags, filt = ss.forward(cloud, cloud, cameras=(views, projs))
        idx, zb = frags.idx, frags.zbuf
        valid = idx >= 0
        assert torch.equal(frags.occupancy.bool(), valid[..., 0])
        assert (valid[..., 1:] <= valid[..., :-1]).all()
        z = torch.where(valid, zb, torch.full_like(zb, float("inf")))
        assert (z[..., 1:] >= z[..., :-1]).all() and ((zb - zb[..., :1])[valid] <= 0.05).all()
        tot = filt["ndc"].shape[0]
        assert int(idx.max()) < tot
        img = composite(frags, filt["scaler"], 0.5 * (filt["normals"] + 1))
        assert torch.isfinite(img).all() and torch.equal(img[..., 3], frags.occupancy)
        occ_grad = 2.0 * (img[..., 3] - 0.5) / img[..., 3].numel()
        zg = torch.zeros_like(zb)
        zg[..., 0] = 1e-3
        vis, rs_ = _visible_and_radius(idx, filt["radii"], filt["first_idx"], filt["num_points"], 10.0)
        grad = _C._backward(filt["ndc"], filt["radii"], occ_grad, filt["first_idx"], filt["num_points"], visible=vis,
                            rs=rs_, idx=idx, grad_zbuf=zg)
        assert torch.isfinite(grad).all()
        # z gradient = 1e-3 x (number of pixels whose first slot lists the point): exact in fixed point
        cnt = torch.bincount(idx[..., 0][valid[..., 0]].long(), minlength=tot).float()
        assert torch.equal(grad[:, 2], cnt * 1e-3) or float((grad[:, 2] - cnt * 1e-3).abs().max()) <= 1e-3 * 2 ** -20 * float(cnt.max())
        assert (grad[~vis.bool()][:, :2] == 0).all()                      # invisible points get no xy gradient
