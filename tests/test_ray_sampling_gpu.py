"""Ray-side sampling around the iso-points (combined_modeling.py:317-386):
fused ray -> nearest-point search vs the dense (R,M) restatement, segment bounds, lowest-SDF
candidate per ray, and the visible-point filter."""
import math

import pytest
import torch

from util import fitted_siren, rel_err, sphere_cloud

pytestmark = pytest.mark.gpu


def _O():
    from oracle import iso_oracle as O
    return O


def _camera_rays(n, seed, cam):
    g = torch.Generator().manual_seed(seed)
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.6
    return torch.nn.functional.normalize(tgt - cam, dim=-1)


@pytest.mark.parametrize("R,M", [(1000, 5000), (37, 1), (4096, 70001), (3, 2500)])
def test_ray_nearest_point_vs_dense(dev, R, M):
    O = _O()
    from iso_points_amd.ray_sampling import ray_nearest_point
    cam = torch.tensor([0.3, 0.5, 2.8])
    rays = _camera_rays(R, 1, cam)
    pts = sphere_cloud(M, seed=2)[0]
    sq_ref, idx_ref, d_ref, dense = O.ray_nearest_point(rays, cam, pts)
    sq, idx, d = ray_nearest_point(rays.to(dev), cam, pts.to(dev))
    idx, sq, d = idx.cpu(), sq.cpu(), d.cpu()
    assert idx.dtype == torch.int64 and int(idx.min()) >= 0 and int(idx.max()) < M
    # the chosen point attains the minimum of the dense row (to rounding: the row itself is only
    # accurate to ~1e-6 absolute, two near-equal candidates may swap)
    chosen = dense[torch.arange(R), idx]
    assert float((chosen - d_ref).abs().max()) < 2e-6
    assert (idx == idx_ref).float().mean() > 0.99
    same = idx == idx_ref
    assert torch.equal(sq[same], sq_ref[same]) and torch.equal(d[same], d_ref[same])       # bit-exact arithmetic


def test_ray_nearest_point_edge_cases(dev):
    from iso_points_amd.ray_sampling import ray_nearest_point
    cam = torch.zeros(3)
    rays = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]], device=dev)
    # no points: index -1, ray_sq 0
    sq, idx, d = ray_nearest_point(rays, cam, torch.zeros(0, 3, device=dev))
    assert idx.tolist() == [-1, -1] and sq.tolist() == [0.0, 0.0]
    # no rays
    sq, idx, d = ray_nearest_point(torch.zeros(0, 3, device=dev), cam, torch.rand(10, 3, device=dev))
    assert sq.shape == (0,) and idx.shape == (0,)
    # ties go to the lowest index; a point behind the camera counts like one in front (reference quirk:
    # only the squared projection is used)
    pts = torch.tensor([[0.0, 0.5, 2.0], [0.0, 0.5, 2.0], [0.0, 0.1, -3.0], [0.5, 0.0, 9.0]], device=dev)
    sq, idx, d = ray_nearest_point(rays, cam, pts)
    assert idx.tolist() == [2, 0]
    assert abs(float(sq[0]) - 9.0) < 1e-6
    with pytest.raises(RuntimeError):
        ray_nearest_point(rays.cpu(), cam, pts)


def test_insurface_segments_and_lowest_sdf(dev):
    """A unit sphere seen from z = +3: frontal points = near hemisphere, occluded = far hemisphere;
    the in-surface segment of a ray through the ball lies inside it and the lowest-SDF candidate of
    the analytic / SIREN sphere sits near the middle of the chord."""
    O = _O()
    from iso_points_amd.ray_sampling import insurface_segments, lowest_sdf_on_segments
    from iso_points_amd.sdf_models import SphereSDF
    cam = torch.tensor([0.0, 0.0, 3.0])
    cloud = torch.nn.functional.normalize(sphere_cloud(40000, seed=5)[0], dim=-1)
    front, back = cloud[cloud[:, 2] > 0.05], cloud[cloud[:, 2] < -0.05]
    g = torch.Generator().manual_seed(9)
    tgt = torch.cat([(torch.rand(3000, 2, generator=g) - 0.5) * 1.2, torch.zeros(3000, 1)], -1)
    rays = torch.nn.functional.normalize(tgt - cam, dim=-1)
    l0, l1, valid = insurface_segments(cam, rays.to(dev), front.to(dev), back.to(dev))
    r0, r1, rv = O.insurface_segments(cam, rays, front, back)
    assert (valid.cpu() == rv).float().mean() > 0.995
    assert rel_err(l0, r0) < 1e-3 and rel_err(l1, r1) < 1e-3          # a swapped near-tie moves the bound slightly
    assert float(valid.float().mean()) > 0.95
    v = valid.cpu()
    # chord of the unit sphere: entry / exit distances along the ray
    b = (rays * cam).sum(-1)
    disc = (b * b - (cam.dot(cam) - 1.0)).clamp_min(0).sqrt()
    t_in, t_out = -b - disc, -b + disc
    assert float((l0.cpu()[v] - t_in[v]).abs().max()) < 0.05 and float((l1.cpu()[v] - t_out[v]).abs().max()) < 0.05
    for model_cpu, model_gpu in ((O.SphereSDF(), SphereSDF().to(dev)),) + tuple(
            (m, m_) for m in [fitted_siren(O, 256, 3, seed=0, fit=200)] for m_ in [__import__("copy").deepcopy(m).to(dev)]):
        cr = rays[v]
        p = lowest_sdf_on_segments(model_gpu, cam.to(dev), cr.to(dev), l0[valid], l1[valid], n_points_per_ray=64)
        p_ref, val_ref = O.lowest_sdf_on_segments(model_cpu, cam, cr, l0.cpu()[v], l1.cpu()[v], n_points_per_ray=64)
        # same candidate unless two candidate values agree to rounding (symmetric chord: i and 63-i)
        same = (p.cpu() - p_ref).abs().amax(-1) < 1e-5
        assert same.float().mean() > 0.9
        val_at = model_cpu.forward(p.cpu()).sdf.view(-1)
        assert float((val_at - val_ref.min(-1).values).detach().abs().max()) < 2e-5    # equally low where it differs
        assert float(p.norm(dim=-1).max()) < 1.0 + 1e-3


def test_get_visible_points(dev):
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.ray_sampling import get_visible_points
    pts = torch.nn.functional.normalize(sphere_cloud(30000, seed=3)[0], dim=-1).to(dev)
    nrm = pts.clone()
    views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(3)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    vis, mask = get_visible_points(pts, nrm, (views, projs), return_mask=True)
    assert len(vis) == 3 and mask.shape == (3, 30000) and mask.dtype == torch.bool
    for i in range(3):
        assert torch.equal(pts[mask[i]], vis[i])
        e, a = math.radians(20.0), math.radians(90.0 * i)
        cam = torch.tensor([3 * math.cos(e) * math.sin(a), 3 * math.sin(e), 3 * math.cos(e) * math.cos(a)], device=dev)
        facing = ((vis[i] - cam) * vis[i]).sum(-1)
        # only front-facing points own a fragment (the culling test is the view-space normal's z, which
        # differs from the exact (p - C).n at the silhouette of a perspective camera)
        assert float((facing < 0).float().mean()) > 0.97
        frac = float(mask[i].float().mean())
        assert 0.1 < frac < 0.5          # the cap seen from distance 3, clipped by the 30 degree frustum
    # the camera looking from the opposite side sees a disjoint set (cameras_back, :321-328)
    views_b = torch.stack([look_at_view(3.0, 20.0, 0.0), look_at_view(3.0, -20.0, 180.0)]).to(dev)
    vis_b, mask_b = get_visible_points(pts, nrm, (views_b, views_b @ perspective(30.0).to(dev)), return_mask=True)
    assert int((mask_b[0] & mask_b[1]).sum()) == 0


def test_get_tensor_values_golden(dev):
    """iso_image_sample vs the reference's own get_tensor_values (tests/golden/make_golden_image.py)."""
    from test_oracle_golden import load
    from iso_points_amd.ray_sampling import get_tensor_values
    g = load("image_values.npz")
    v, m = get_tensor_values(g["mask"].to(dev), g["p"].to(dev), with_mask=True, squeeze_channel_dim=True)
    assert v.shape == g["mask_bilinear"].shape and m.dtype == torch.bool
    assert (v.cpu() - g["mask_bilinear"]).abs().max() < 1e-6 and torch.equal(m.cpu(), g["mask_valid"])
    out = get_tensor_values(g["rgb"].to(dev), g["p"].to(dev))
    assert out.shape == g["rgb_bilinear"].shape and (out.cpu() - g["rgb_bilinear"]).abs().max() < 1e-6
    near = get_tensor_values(g["rgb"].to(dev), g["p"].to(dev), mode="nearest").cpu()
    assert (near != g["rgb_nearest"]).any(-1).float().mean() < 2e-3     # x.5 ties after float rounding
    p_in = g["p_in"].to(dev)
    keep = p_in.clone()
    assert torch.equal(get_tensor_values(g["sq"].to(dev), p_in, grid_sample=False).cpu(), g["sq_index"])
    assert torch.equal(p_in, keep)
    with pytest.raises(IndexError):
        get_tensor_values(g["sq"].to(dev), p_in * 3, grid_sample=False)
    with pytest.raises(RuntimeError):
        get_tensor_values(g["sq"], g["p_in"])


def test_get_tensor_values_vs_oracle_large(dev):
    """a 512 x 512 mask sampled at 1 M points (the look-up of every iso-point of cfg 3), bool source"""
    from oracle import iso_oracle as O
    from iso_points_amd.ray_sampling import get_tensor_values
    gen = torch.Generator().manual_seed(8)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
    mask = ((xx * xx + yy * yy) < 0.6).view(1, 1, 512, 512).repeat(4, 1, 1, 1)
    p = (torch.rand(4, 250000, 2, generator=gen) - 0.5) * 2.2
    ref = O.get_tensor_values(mask.float(), p, squeeze_channel_dim=True)
    got = get_tensor_values(mask.to(dev).float(), p.to(dev), squeeze_channel_dim=True)
    assert got.shape == (4, 250000) and (got.cpu() - ref).abs().max() < 2e-5
    empty = get_tensor_values(mask.to(dev).float(), torch.zeros(4, 0, 2, device=dev))
    assert empty.shape == (4, 0, 1)


def test_ray_side_sampling_vs_the_reference_statements(dev):
    """iso_ray_nearest_point + segment bounds + lowest-SDF candidate against the reference's own dense
    (R,M) statements (combined_modeling.py:324-386; tests/golden/ray_sampling.npz)."""
    from test_oracle_golden import load
    from iso_points_amd.ray_sampling import insurface_segments, lowest_sdf_on_segments
    from iso_points_amd.sdf_models import SphereSDF
    g = load("ray_sampling.npz")
    B = g["cam_pos"].shape[0]
    model = SphereSDF(radius=float(g["sdf_radius"])).to(dev)
    l0s, l1s, ps, valid_all = [], [], [], []
    for b in range(B):
        cam = g["cam_pos"][b]
        ray0 = torch.nn.functional.normalize(g["samples"][b] - cam.view(1, 3), dim=-1).to(dev)
        l0, l1, valid = insurface_segments(cam, ray0, g["frontal%d" % b].to(dev), g["occluded%d" % b].to(dev))
        valid_all.append(valid.cpu())
        l0s.append(l0[valid].cpu()); l1s.append(l1[valid].cpu())
        ps.append(lowest_sdf_on_segments(model, cam.to(dev), ray0[valid], l0[valid], l1[valid],
                                         n_points_per_ray=int(g["n_points_per_ray"])).cpu())
    assert torch.equal(torch.stack(valid_all), g["mask_insurface"].bool())
    # the point nearest to a ray is the same unless two distances agree to rounding (the dense statement
    # subtracts two large squares); the bounds then differ by the spacing of the cloud
    e0 = (torch.cat(l0s) - g["ray_len0"]).abs() / g["ray_len0"].abs().max()
    e1 = (torch.cat(l1s) - g["ray_len1"]).abs() / g["ray_len1"].abs().max()
    assert (e0 > 1e-5).float().mean() < 0.01 and (e1 > 1e-5).float().mean() < 0.01
    ok = (e0 <= 1e-5) & (e1 <= 1e-5)
    ep = (torch.cat(ps) - g["p_insurface"]).abs().amax(-1)
    assert (ep[ok] > 1e-5).float().mean() < 0.02       # equal candidate values on a symmetric chord may tie
