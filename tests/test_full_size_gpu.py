"""BASELINE.json full size (configs[2]: 1 M points): the oracle cannot run here in seconds, so the
point-side stages are pinned through size-independent properties of the domain -- idempotence of
the projection, the level-set condition checked by an independent evaluation, sortedness /
self-match / exactness-on-a-sample of the neighbour search, tangent-plane repulsion -- next to
the small-size oracle parity of the other files.  (tests/test_splat_gpu.py::test_full_size_properties
does the same for the raster.)"""
import pytest
import torch

from util import sphere_cloud, fitted_siren

pytestmark = pytest.mark.gpu
P_FULL = 1000000
TOL = 5e-5


def test_full_size_sphere_projection_is_idempotent(dev):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    pts = sphere_cloud(P_FULL, seed=11).to(dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=TOL, knn_k=8)
    m = SphereSDF().to(dev)
    r = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    assert bool(r.mask.all())
    rad = r.points.norm(dim=-1)
    assert float((rad - 1).abs().max()) <= TOL * 1.001
    n_ref = torch.nn.functional.normalize(r.points, dim=-1)
    assert float((r.normals - n_ref).abs().max()) < 2e-6
    r2 = proj._project_points(m, r.points, full_lengths(pts), proj_max_iters=10)
    assert torch.equal(r2.points, r.points) and bool(r2.mask.all())      # a converged cloud does not move


def test_full_size_siren_projection_reaches_the_level_set(dev):
    """|sdf| <= tol on every point flagged converged (checked by an independent fused evaluation),
    almost all points converge on the sphere-fitted network, the returned normals are the
    gradient at the returned points, and projecting the result again does not move it."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import iso_oracle as O
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    m = fitted_siren(O, 256, 3, seed=0, fit=200).to(dev)
    pts = sphere_cloud(P_FULL, seed=12).to(dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=TOL, knn_k=8)
    r = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    mask = r.mask[0].bool()
    assert float(mask.float().mean()) > 0.98
    sdf, grad = siren_sdf_and_grad(m, r.points[0])
    assert float(sdf[mask].abs().max()) <= TOL                         # same kernel, same point -> same value
    assert torch.equal(grad[mask], r.normals[0][mask])
    r2 = proj._project_points(m, r.points, full_lengths(pts), proj_max_iters=10)
    assert torch.equal(r2.points[0][mask], r.points[0][mask])


@pytest.mark.parametrize("P_N", [P_FULL, 400000])       # dense-grid caps 256 and 192 (frnn.grid_max_res)
def test_full_size_neighbour_search(dev, P_N):
    from iso_points_amd import frnn
    from iso_points_amd.levelset_sampling import cloud_diag, full_lengths
    K = 9
    P_FULL = P_N
    pts = torch.nn.functional.normalize(sphere_cloud(P_FULL, seed=13), dim=-1).to(dev).contiguous()
    num = full_lengths(pts)
    radius = (torch.sqrt(cloud_diag(pts) / num.float()) * 8).contiguous()      # levelset_sampling.py:129-131
    dists, idxs, nn, grid = frnn.frnn_grid_points(pts, pts, num, num, K=K, r=radius, return_nn=True)
    d, i = dists[0], idxs[0]
    rows = torch.arange(P_FULL, device=dev)
    assert torch.equal(i[:, 0], rows) and float(d[:, 0].abs().max()) == 0.0     # self first, distance 0
    found = i >= 0
    assert bool((d[found] < radius[0] ** 2).all())
    dd = torch.where(found, d, torch.full_like(d, float("inf")))
    assert bool((dd[:, 1:] >= dd[:, :-1]).all())                                # ascending, padding last
    assert bool((found[:, 1:] <= found[:, :-1]).all())                          # -1 padding is a suffix
    srt = torch.where(found, i, -1 - torch.arange(K, device=dev)[None]).sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                              # no duplicates in a row
    assert torch.equal(nn[0][found], pts[0][i[found]])                          # nn = gathered positions
    # exactness on a sample: brute force over the whole cloud with the kernel's own formula
    g = torch.Generator().manual_seed(3)
    sample = torch.randint(0, P_FULL, (256,), generator=g).to(dev)
    q = pts[0][sample]
    dx = q[:, None, 0] - pts[0][None, :, 0]
    dy = q[:, None, 1] - pts[0][None, :, 1]
    dz = q[:, None, 2] - pts[0][None, :, 2]
    d2 = (dx * dx + dy * dy) + dz * dz
    d2 = torch.where(d2 < radius[0] ** 2, d2, torch.full_like(d2, float("inf")))
    # K smallest by (d2, index): ties are broken by the lower index (stable sort of the index-ordered row)
    vals, order = torch.sort(d2, dim=1, stable=True)
    bd, bi = vals[:, :K], order[:, :K]
    ok = torch.isfinite(bd)
    assert torch.equal(torch.where(ok, bi, torch.full_like(bi, -1)), i[sample])
    assert torch.equal(torch.where(ok, bd, torch.full_like(bd, -1.0)), d[sample])


def test_full_size_resample_keeps_the_surface_and_spreads_the_points(dev):
    from iso_points_amd import frnn
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    m = SphereSDF().to(dev)
    pts = sphere_cloud(P_FULL, seed=14).to(dev)
    num = full_lengths(pts)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=TOL, knn_k=8, sample_iters=1)
    r0 = proj._project_points(m, pts, num, proj_max_iters=10)
    r1 = proj.resample(m, r0.points, r0.normals, num, sample_iters=1)
    assert bool(torch.isfinite(r1.points).all())
    assert float((r1.points.norm(dim=-1) - 1).abs().max()) <= TOL * 1.001      # back on the level set
    # the repulsion evens the sampling out: the nearest-neighbour distance distribution tightens

    def nn_dist(p):
        d, _, _, _ = frnn.frnn_grid_points(p, p, num, num, K=2, r=0.1)
        return d[0][:, 1].clamp_min(0).sqrt()
    before, after = nn_dist(r0.points.contiguous()), nn_dist(r1.points.contiguous())
    assert float(after.mean()) > float(before.mean())
    assert float(after.quantile(0.01)) > float(before.quantile(0.01))
