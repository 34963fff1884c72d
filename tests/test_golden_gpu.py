"""HIP path vs the golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  Same tolerances as the oracle comparisons."""
import pytest
import torch

from test_oracle_golden import load, siren_from
from util import assert_projection_close, rel_err

pytestmark = pytest.mark.gpu


def test_cfg1_projection_sphere(dev):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    g = load("proj_sphere_cfg1.npz")
    x = g["points"].to(dev)
    for T in (1, 10):
        r = UniformProjection()._project_points(SphereSDF().to(dev), x, full_lengths(x), proj_max_iters=T)
        assert_projection_close(r.points, g["T%d_points" % T])
        assert (r.mask.cpu() == g["T%d_mask" % T]).float().mean() > 0.999


def test_ragged_projection(dev):
    from iso_points_amd.levelset_sampling import UniformProjection
    from iso_points_amd.sdf_models import SphereSDF
    g = load("proj_sphere_ragged.npz")
    m = SphereSDF(tuple(g["center"].tolist()), float(g["radius"])).to(dev)
    r = UniformProjection()._project_points(m, g["points"].to(dev), g["num_points"].to(dev),
                                            proj_max_iters=int(g["T"]))
    assert r.points.shape == g["out_points"].shape
    assert_projection_close(r.points, g["out_points"])
    assert (r.mask.cpu() == g["out_mask"]).float().mean() > 0.999


def test_siren_eval_and_fixed_count_projection(dev):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    g = load("proj_siren_small.npz")
    m = siren_from(g)
    sdf, grad = siren_sdf_and_grad(m, g["points"].to(dev))
    assert rel_err(sdf, g["sdf"]) < 1e-5 and rel_err(grad, g["grad"]) < 1e-5
    g = load("proj_siren_fitted.npz")
    m = siren_from(g)
    x = g["points"].to(dev)
    r0 = UniformProjection(proj_tolerance=1e-30)._project_points(m, x, full_lengths(x), proj_max_iters=10)
    assert rel_err(r0.points, g["fixed_points"]) < 1e-5
    # gradients after ten moves: the yardstick is a float64 iteration of the same network.  The reference's OWN float32
    # result (the golden) is up to 1.4e-5 away from it (measured: tools/diag/tolerance_probe.py), so "1e-5 against the
    # golden" cannot be asked of anybody; what is asked: within 1e-5 of the float64 truth on EVERY point (measured
    # 3.8e-6), no further from it than 1.5 x the reference's own float32 error quantile by quantile (measured ratios
    # 0.28-1.02 split16, 0.28-1.23 f32 MFMA), and within 1e-5 + the golden's own error of the golden
    import copy
    from oracle import iso_oracle as O
    r64 = O.project_points(copy.deepcopy(m).cpu().double(), g["points"].double(), torch.tensor([g["points"].shape[1]]),
                           proj_max_iters=10, proj_tolerance=1e-30)
    scale = r64.normals.abs().max()
    e_ref = ((g["fixed_normals"].double() - r64.normals).abs().amax(-1) / scale).view(-1)
    e_our = ((r0.normals.cpu().double() - r64.normals).abs().amax(-1) / scale).view(-1)
    assert e_our.max() < 1e-5, e_our.max().item()
    assert rel_err(r0.normals, g["fixed_normals"]) < 1e-5 + e_ref.max().item()
    for q in (0.5, 0.9, 0.99, 1.0):
        assert torch.quantile(e_our, q) <= 1.5 * torch.quantile(e_ref, q) + 2e-7, (q, torch.quantile(e_our, q).item(),
                                                                                  torch.quantile(e_ref, q).item())
    r = UniformProjection()._project_points(m, x, full_lengths(x), proj_max_iters=10)
    assert_projection_close(r.points, g["out_points"])


@pytest.mark.parametrize("tag", ["sphere", "sphere3", "siren"])
def test_resample(dev, tag):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    g = load("resample_%s.npz" % tag)
    m = siren_from(g) if tag == "siren" else SphereSDF().to(dev)
    pp = g["proj_points"].to(dev)
    up = UniformProjection(knn_k=int(g["knn_k"]))
    r = up.resample(m, pp, g["proj_normals"].to(dev), full_lengths(pp), sample_iters=int(g["sample_iters"]))
    assert_projection_close(r.points, g["out_points"])
    assert (r.mask.cpu() == g["out_mask"]).float().mean() > 0.995
    # the tree the reference run cached (built by the brute-force frnn shim) == ours, bit for bit
    if int(g["sample_iters"]) == 1:
        assert torch.equal(up._knn_idx.cpu(), g["knn_idx"])


def test_project_points_driver(dev):
    from iso_points_amd.levelset_sampling import UniformProjection
    from iso_points_amd.sdf_models import SphereSDF
    g = load("project_points_driver.npz")
    out = UniformProjection(proj_max_iters=int(g["T"]), knn_k=int(g["knn_k"])).project_points(
        g["points"].to(dev), SphereSDF().to(dev), skip_upsampling=True)
    assert out["levelset_points"].shape == g["levelset_points"].shape
    assert_projection_close(out["levelset_points"], g["levelset_points"])


@pytest.mark.parametrize("name", ["siren_ref_128x2.npz", "siren_ref_256x4.npz"])
@pytest.mark.parametrize("mode", ["split16", "f32"])
def test_fused_siren_vs_the_reference_siren_class(dev, name, mode):
    """the fused SDF + gradient kernel and the Newton projection on the weights of the
    reference's own Siren class (common.py:90-165) against what the reference computed with them
    (model.forward + autograd, 4 clamped Newton moves), 1e-5 relative."""
    from test_oracle_golden import load, siren_from_ref
    from iso_points_amd import _lib
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    from util import rel_err
    g = load(name)
    m = siren_from_ref(g).to(dev)
    lib = _lib.load()
    lib.iso_siren_set_gemm_mode(1 if mode == "split16" else 0)
    try:
        sdf, grad = siren_sdf_and_grad(m, g["points"].to(dev))
        assert rel_err(sdf, g["sdf"].reshape(sdf.shape)) < 1e-5 and rel_err(grad, g["grad"].reshape(grad.shape)) < 1e-5
        x = g["points"].to(dev)
        r = UniformProjection(proj_tolerance=1e-30)._project_points(m, x, full_lengths(x), proj_max_iters=int(g["T"]))
        # A randomly initialised SIREN (omega = 30) is not an SDF: four clamped Newton moves on it amplify
        # rounding differences by orders of magnitude on some points.  The yardstick is therefore float64: the
        # reference's own float32 result (the golden) deviates from the float64 iteration by e_ref; the fused
        # kernel must deviate by no more than that, quantile by quantile (and agree with the golden to 1e-5
        # on the bulk of the points).
        from oracle import iso_oracle as O
        import copy
        m64 = copy.deepcopy(siren_from_ref(g)).double()
        r64 = O.project_points(m64, g["points"].double(), torch.tensor([g["points"].shape[1]]),
                               proj_max_iters=int(g["T"]), proj_tolerance=1e-30)
        scale = r64.points.abs().max()
        e_ref = ((g["fixed_points"].double() - r64.points).abs().amax(-1) / scale).view(-1)
        e_our = ((r.points.cpu().double() - r64.points).abs().amax(-1) / scale).view(-1)
        # measured ratios (tools/diag/tolerance_probe.py): split16 1.01-1.22, f32 MFMA 0.36-1.46
        for q in (0.5, 0.9, 0.99, 1.0):
            assert torch.quantile(e_our, q) <= 1.5 * torch.quantile(e_ref, q) + 2e-7, (q, torch.quantile(e_our, q), torch.quantile(e_ref, q))
        e_g = ((r.points.cpu() - g["fixed_points"]).abs().amax(-1) / g["fixed_points"].abs().max()).view(-1)
        # against the golden itself: the chaotic points (both float32 runs far from float64) are 0.07 % / 0.40 % of the cloud
        assert (e_g > 1e-5).float().mean() < 0.006 and e_g.median() < 1e-6
    finally:
        lib.iso_siren_set_gemm_mode(1)
