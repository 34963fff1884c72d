"""IDR's two-ended ray tracer on the fused value-only SDF kernels: against the
golden vectors made by the reference's own RayTracing.forward (tests/golden/make_golden_raytrace.py)
and against the oracle's restatement on seeded inputs, in eval and in training mode."""
import pytest
import torch

from test_oracle_golden import RT_CASES, RT_CASES_SIREN, assert_raytrace_close, load, siren_from

pytestmark = pytest.mark.gpu


def _O():
    from oracle import iso_oracle as O
    return O


def _trace(dev, fused, g, training, kw, uniform):
    from iso_points_amd.ray_tracing import RayTracing
    rt = RayTracing(**kw).train(training)
    return rt(sdf=fused, cam_loc=g["cam"].to(dev), object_mask=g["gt"].to(dev),
              ray_directions=g["dirs"].to(dev), uniform_steps=uniform)


def test_ray_tracing_sphere_golden(dev):
    from iso_points_amd.sdf_models import FusedSdf, SphereSDF
    g = load("raytrace_sphere.npz")
    fused = FusedSdf(SphereSDF(tuple(g["center"].tolist()), float(g["radius"])).to(dev), dev)
    for tag, (training, kw) in RT_CASES.items():
        got = _trace(dev, fused, g, training, kw, g[tag + "_uniform"])
        assert got[0].shape == g[tag + "_points"].shape and got[1].dtype == torch.bool
        # sqrt / norm round differently on the two devices; a ray may flip at the 5e-5 threshold
        assert_raytrace_close(got, g, tag, tol=2e-6, flip=2e-3, frac=0.995)


@pytest.mark.parametrize("mode", ["split16", "f32"])
def test_ray_tracing_siren_golden(dev, mode):
    from iso_points_amd import _lib
    from iso_points_amd.sdf_models import FusedSdf
    g = load("raytrace_siren.npz")
    net = siren_from(load("trace_siren.npz")).to(dev)
    old = _lib.load().iso_siren_get_gemm_mode()
    _lib.call("iso_siren_set_gemm_mode", 1 if mode == "split16" else 0)
    try:
        fused = FusedSdf(net, dev)
        for tag, (training, kw) in RT_CASES_SIREN.items():
            got = _trace(dev, fused, g, training, kw, g[tag + "_uniform"])
            assert_raytrace_close(got, g, tag, tol=1e-5, flip=5e-3, frac=0.99)
    finally:
        _lib.call("iso_siren_set_gemm_mode", old)


def _pixel_rays(n, seed, cam, spread):
    gen = torch.Generator().manual_seed(seed)
    c = torch.tensor(cam)
    tgt = (torch.rand(n, 3, generator=gen) - 0.5) * 2 * spread
    return c.view(1, 3), torch.nn.functional.normalize(tgt - c, dim=-1).view(1, n, 3)


@pytest.mark.parametrize("training", [False, True])
def test_ray_tracing_idr_vs_oracle(dev, training):
    """geometric-init IDR network (~ sphere of radius 0.6), two cameras in one batch."""
    O = _O()
    from iso_points_amd.ray_tracing import RayTracing
    from iso_points_amd.sdf_models import FusedSdf
    torch.manual_seed(5)
    net = O.IdrSDF(hidden_size=256, n_layers=4, skip_in=(2,), num_frequencies=4)
    c0, d0 = _pixel_rays(1200, 31, (0.0, 0.2, 2.2), 1.1)
    c1, d1 = _pixel_rays(1200, 32, (-1.5, 1.0, 1.0), 1.1)
    cam, dirs = torch.cat([c0, c1]), torch.cat([d0, d1])
    gt = torch.rand(2400, generator=torch.Generator().manual_seed(33)) > 0.4
    u = torch.rand(50, generator=torch.Generator().manual_seed(34))
    kw = {"n_steps": 50, "sphere_tracing_iters": 6, "line_step_iters": 2}
    ref = O.ray_tracing(lambda x: net.forward(x).sdf.reshape(-1), cam, gt, dirs, training=training,
                        uniform_steps=u, **kw)
    rt = RayTracing(**kw).train(training)
    got = rt(sdf=FusedSdf(net.to(dev), dev), cam_loc=cam.to(dev), object_mask=gt.to(dev),
             ray_directions=dirs.to(dev), uniform_steps=u)
    g = {"x_points": ref[0], "x_mask": ref[1], "x_dist": ref[2]}
    assert 0.1 < ref[1].float().mean() < 0.9
    assert_raytrace_close(got, g, "x", tol=1e-5, flip=5e-3, frac=0.99)


def test_ray_tracing_edge_cases(dev):
    from iso_points_amd.ray_tracing import RayTracing
    from iso_points_amd.sdf_models import FusedSdf, SphereSDF
    O = _O()
    fused = FusedSdf(SphereSDF((0.0, 0.0, 0.0), 0.5).to(dev), dev)
    cam = torch.tensor([[0.0, 0.0, -3.0]])
    # centre hit, grazing miss of the shape (inside the bounding sphere), miss of the bounding sphere
    dirs = torch.nn.functional.normalize(torch.tensor([[[0.0, 0.0, 1.0], [0.0, 0.25, 1.0], [0.0, 0.6, 1.0]]]), dim=-1)
    gt = torch.tensor([True, True, False])
    for training in (False, True):
        rt = RayTracing().train(training)
        u = torch.linspace(0.05, 0.95, 100)
        pts, mask, z = rt(sdf=fused, cam_loc=cam.to(dev), object_mask=gt.to(dev), ray_directions=dirs.to(dev),
                          uniform_steps=u)
        ref = O.ray_tracing(lambda x: O.SphereSDF(radius=0.5).forward(x).sdf.reshape(-1), cam, gt, dirs,
                            training=training, uniform_steps=u)
        assert mask.tolist() == [True, False, False] == ref[1].tolist()
        assert (pts.cpu() - ref[0]).abs().max() < 5e-6 and (z.cpu() - ref[2]).abs().max() < 5e-6
        assert abs(z[0].item() - 2.5) < 1e-4
    # the default draw of the minimal-value depths comes from the CPU generator (:1142)
    rt = RayTracing().train(True)
    torch.manual_seed(9)
    a = rt(sdf=fused, cam_loc=cam.to(dev), object_mask=gt.to(dev), ray_directions=dirs.to(dev))
    torch.manual_seed(9)
    b = rt(sdf=fused, cam_loc=cam.to(dev), object_mask=gt.to(dev), ray_directions=dirs.to(dev),
           uniform_steps=torch.empty(100).uniform_(0.0, 1.0))
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    with pytest.raises(RuntimeError):
        rt(sdf=fused, cam_loc=cam, object_mask=gt, ray_directions=dirs)
    # no rays; rays that all miss the bounding sphere; a camera inside the bounding sphere
    for training in (False, True):
        rt = RayTracing().train(training)
        pts, mask, z = rt(sdf=fused, cam_loc=cam.to(dev), object_mask=gt[:0].to(dev),
                          ray_directions=dirs[:, :0].to(dev))
        assert pts.shape == (0, 3) and mask.shape == (0,) and z.shape == (0,)
        away = torch.nn.functional.normalize(torch.tensor([[[0.0, 1.0, 0.2], [1.0, 0.0, 0.1]]]), dim=-1)
        u = torch.linspace(0.05, 0.95, 100)
        for c, d, m in ((cam, away, torch.tensor([True, False])),
                        (torch.tensor([[0.1, 0.0, -0.8]]), dirs, gt)):
            got = rt(sdf=fused, cam_loc=c.to(dev), object_mask=m.to(dev), ray_directions=d.to(dev), uniform_steps=u)
            ref = O.ray_tracing(lambda x: O.SphereSDF(radius=0.5).forward(x).sdf.reshape(-1), c, m, d,
                                training=training, uniform_steps=u)
            assert got[1].tolist() == ref[1].tolist()
            assert (got[0].cpu() - ref[0]).abs().max() < 5e-6 and (got[2].cpu() - ref[2]).abs().max() < 5e-6
