"""HIP Newton projection vs the oracle (UniformProjection._project_points,
levelset_sampling.py:290-351).  Tolerance: 1e-5 relative on positions / gradients
(BASELINE.json north_star), masks equal except where |sdf| sits within 1e-6 of tol."""
import pytest
import torch

from util import cube_cloud, sphere_cloud, rel_err, assert_projection_close

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle():
    from oracle import iso_oracle
    return iso_oracle


def _check_projection(res, ref, model_cpu, tol_sdf=5e-5):
    assert res.points.shape == ref.points.shape
    assert_projection_close(res.points, ref.points, tol_sdf)
    nscale = ref.normals.abs().max()
    nerr = (res.normals.cpu() - ref.normals).abs().amax(-1) / nscale
    assert (nerr > 5 * TOL).float().mean() < 5e-3
    mism = (res.mask.cpu() != ref.mask)
    if mism.any():  # only borderline points may flip
        sdf = model_cpu(ref.points[mism]).sdf.abs().reshape(-1)
        assert ((sdf - tol_sdf).abs() < 2e-6).all(), "mask mismatch away from the tolerance"
        assert mism.float().mean() < 1e-3


@pytest.mark.parametrize("T", [1, 10])
def test_project_sphere_cfg1(dev, T):
    """BASELINE.json configs[0]: 10k points, analytic unit sphere."""
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    pts = cube_cloud(10000, seed=0)
    ref = O.project_points(O.SphereSDF(), pts, torch.tensor([10000]), proj_max_iters=T)
    proj = UniformProjection(proj_max_iters=T)
    g = pts.to(dev)
    res = proj._project_points(SphereSDF().to(dev), g, full_lengths(g), proj_max_iters=T)
    _check_projection(res, ref, O.SphereSDF())
    if T == 1:
        frac = res.mask.float().mean().item()
        assert 0.2 < frac < 0.4  # ~29 % converge after one clamped step (SURVEY 8(d) cfg 1)
    else:
        assert res.mask.all()
        assert (res.points.norm(dim=-1) - 1).abs().max() < 1e-4


def test_project_sphere_ragged_batch_and_offset_center(dev):
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection
    from iso_points_amd.sdf_models import SphereSDF
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(3, 700, 3, generator=g) - 0.5) * 2
    num = torch.tensor([700, 1, 333])
    ref = O.project_points(O.SphereSDF((0.1, -0.2, 0.05), 0.7), pts, num, proj_max_iters=6)
    res = UniformProjection()._project_points(SphereSDF((0.1, -0.2, 0.05), 0.7).to(dev), pts.to(dev),
                                              num.to(dev), proj_max_iters=6)
    assert res.points.shape == (3, 700, 3)
    _check_projection(res, ref, O.SphereSDF((0.1, -0.2, 0.05), 0.7))
    assert (res.points[1, 1:] == 0).all() and not res.mask[1, 1:].any()  # padding stays 0 / False


def test_project_empty(dev):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    g = torch.zeros(1, 0, 3, device=dev)
    res = UniformProjection()._project_points(SphereSDF().to(dev), g, full_lengths(g))
    assert res.points.shape == (1, 0, 3) and res.mask.shape == (1, 0)


def _siren(hidden, n_layers, seed=0, fit=0):
    from util import fitted_siren
    return fitted_siren(_oracle(), hidden, n_layers, seed=seed, fit=fit)


@pytest.mark.parametrize("hidden,n_layers", [(256, 3), (64, 1), (128, 2), (256, 0), (256, 1), (128, 4), (96, 2), (200, 3),
                                              (32, 1)])
def test_siren_sdf_and_grad(dev, hidden, n_layers, gemm_mode):
    """Fused MFMA SDF+grad vs torch autograd (levelset_sampling.py:142-170); widths other than 64 / 128 / 256 run
    zero-padded to the next fused width (PackedSiren) -- same function, same gradient."""
    O = _oracle()
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    m = _siren(hidden, n_layers, seed=1)
    pts = cube_cloud(3001, seed=5)[0]
    sdf_ref, grad_ref = O.compute_sdf_and_grad(pts, m)
    sdf, grad = siren_sdf_and_grad(m, pts.to(dev))
    assert rel_err(sdf, sdf_ref) < TOL
    assert rel_err(grad, grad_ref) < TOL


@pytest.fixture(params=["split16", "f32"])
def gemm_mode(request):
    """Both ways of forming the hidden-layer products (include/isopoints.h: iso_siren_set_gemm_mode)."""
    from iso_points_amd import _lib
    lib = _lib.load()
    before = lib.iso_siren_get_gemm_mode()
    _lib.call("iso_siren_set_gemm_mode", 1 if request.param == "split16" else 0)
    yield request.param
    _lib.call("iso_siren_set_gemm_mode", before)


def test_siren_grad_accuracy_vs_float64(dev, gemm_mode):
    """The fused kernel's float32 SDF/gradient is as close to the float64 value as torch's
    float32 autograd (the reference path) is -- for the f32 matrix cores and for the split-fp16
    products ("split16": two fp16 parts per operand under exact power-of-two scales) alike."""
    import copy
    O = _oracle()
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    m = _siren(256, 3, seed=0, fit=200)
    pts = sphere_cloud(5000, seed=3)[0]
    m64 = copy.deepcopy(m).double()
    sdf64, grad64 = O.compute_sdf_and_grad(pts.double(), m64)
    sdf32, grad32 = O.compute_sdf_and_grad(pts, m)
    sdf, grad = siren_sdf_and_grad(m, pts.to(dev))
    e_ref = (grad32.double() - grad64).abs().max().item()
    e_hip = (grad.cpu().double() - grad64).abs().max().item()
    m_ref = (grad32.double() - grad64).abs().mean().item()
    m_hip = (grad.cpu().double() - grad64).abs().mean().item()
    print("[%s] grad err vs f64: max torch-f32 %.3g hip %.3g | mean torch-f32 %.3g hip %.3g"
          % (gemm_mode, e_ref, e_hip, m_ref, m_hip))
    assert m_hip <= 1.5 * m_ref + 1e-8
    assert e_hip <= 2.0 * e_ref + 1e-6
    s_ref = (sdf32.double() - sdf64).abs().max().item()
    s_hip = (sdf.cpu().double() - sdf64).abs().max().item()
    assert s_hip <= 2.0 * s_ref + 1e-7


@pytest.mark.parametrize("w_scale,head_scale,in_scale", [(1.0, 1.0, 1.0), (40.0, 1.0, 1.0), (1.0, 3.0e4, 1.0),
                                                        (1.0, 1.0e-9, 1.0), (0.02, 1.0, 1.0), (300.0, 1.0e-3, 5.0)])
@pytest.mark.parametrize("hidden,n_layers", [(256, 3), (128, 2)])
def test_split16_operand_ranges(dev, hidden, n_layers, w_scale, head_scale, in_scale):
    """fp16 has a 5-bit exponent: the split-fp16 kernel relies on exact power-of-two scales (per layer
    for the weights, 2^12 for the activations, per point for the adjoint).  Hidden weights 40x / 300x
    larger or 50x smaller than the SIREN initialisation, a head that makes the gradient 3e4 or 1e-9,
    inputs far outside the unit cube: value and gradient must stay finite and within 1e-5 (relative to
    the largest component) of the float32 oracle, exactly as with the f32 matrix cores."""
    O = _oracle()
    from iso_points_amd import _lib
    from iso_points_amd.sdf_models import siren_sdf_and_grad
    torch.manual_seed(hidden + n_layers)
    m = O.SirenSDF(hidden_size=hidden, n_layers=n_layers)
    with torch.no_grad():
        for lin in m.lins[1:-1]:
            lin.weight.mul_(w_scale)
        m.lins[-1].weight.mul_(head_scale)
        m.lins[-1].bias.mul_(head_scale)
    pts = cube_cloud(3000, seed=4)[0] * in_scale
    m64 = __import__("copy").deepcopy(m).double()
    sdf64, grad64 = O.compute_sdf_and_grad(pts.double(), m64)
    sdf32, grad32 = O.compute_sdf_and_grad(pts, m)
    out = {}
    lib = _lib.load()
    before = lib.iso_siren_get_gemm_mode()
    try:
        for mode in (1, 0):
            _lib.call("iso_siren_set_gemm_mode", mode)
            out[mode] = siren_sdf_and_grad(m.to(dev), pts.to(dev))
    finally:
        _lib.call("iso_siren_set_gemm_mode", before)
    gs = grad64.abs().max().item()
    for mode in (1, 0):
        sdf, grad = out[mode]
        assert bool(torch.isfinite(sdf).all()) and bool(torch.isfinite(grad).all())
    # larger weights make the network chaotic (errors of ANY f32 evaluation grow): the fused kernels are
    # held to the error torch's own f32 path makes against float64, not to an absolute number
    e_ref = (grad32.double() - grad64).abs().max().item() / gs
    for mode in (1, 0):
        e = (out[mode][1].cpu().double() - grad64).abs().max().item() / gs
        # (operand-range stress test: scales pushed to their limits; a max-over-points statistic of two float32 paths.
        # The parity tests proper hold the kernels to 1.5 x the reference's own error, quantile by quantile.)
        assert e <= 3.0 * e_ref + 2e-6, (mode, e, e_ref)
    s_ref = (sdf32.double() - sdf64).abs().max().item()
    for mode in (1, 0):
        s_err = (out[mode][0].cpu().double() - sdf64).abs().max().item()
        assert s_err <= 3.0 * s_ref + 1e-6 * sdf64.abs().max().item(), (mode, s_err, s_ref)


def test_siren_reference_layout_is_recognised(dev):
    """A model laid out like the reference's Siren (net[i].linear / omega_0) takes the fused path."""
    O = _oracle()
    from iso_points_amd.sdf_models import Siren, siren_sdf_and_grad, siren_spec
    torch.manual_seed(2)
    m = Siren(dim=3, hidden_size=128, n_layers=2, c_dim=0)
    assert siren_spec(m) is not None
    pts = cube_cloud(500, seed=6)[0]
    sdf_ref, grad_ref = O.compute_sdf_and_grad(pts, m)
    sdf, grad = siren_sdf_and_grad(m.to(dev), pts.to(dev))
    assert rel_err(sdf, sdf_ref) < TOL and rel_err(grad, grad_ref) < TOL


def test_project_siren_fixed_iteration_count_strict(dev, gemm_mode):
    """north_star bar: positions and gradients within 1e-5 relative after a FIXED iteration
    count -- tolerance ~0 so every point takes all T moves on both sides."""
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    m = _siren(256, 3, seed=0, fit=200)
    pts = sphere_cloud(4000, seed=9)
    for T in (1, 3, 10):
        ref = O.project_points(m, pts, torch.tensor([4000]), proj_max_iters=T, proj_tolerance=1e-30)
        g = pts.to(dev)
        res = UniformProjection(proj_tolerance=1e-30)._project_points(m, g, full_lengths(g), proj_max_iters=T)
        assert rel_err(res.points, ref.points) < TOL
        assert rel_err(res.normals, ref.normals) < TOL
        assert not res.mask.any()


@pytest.mark.parametrize("T", [1, 10])
def test_project_siren_fitted(dev, T):
    """cfg 2 shape at test size: SIREN 4x256 fitted to the sphere, jittered sphere samples."""
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    m = _siren(256, 3, seed=0, fit=200)
    pts = sphere_cloud(4000, seed=1)
    ref = O.project_points(m, pts, torch.tensor([4000]), proj_max_iters=T)
    g = pts.to(dev)
    res = UniformProjection()._project_points(m, g, full_lengths(g), proj_max_iters=T)
    _check_projection(res, ref, m)


def test_project_siren_random_weights_fixed_iterations(dev, gemm_mode):
    """Random SIREN (not an SDF): nothing converges, every point takes all T clamped moves --
    exercises the active-list ping-pong for the full iteration count."""
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    m = _siren(256, 3, seed=4)
    pts = sphere_cloud(1500, seed=2)
    ref = O.project_points(m, pts, torch.tensor([1500]), proj_max_iters=4)
    g = pts.to(dev)
    res = UniformProjection()._project_points(m, g, full_lengths(g), proj_max_iters=4)
    # chaotic map: compare with a looser bound, but still far below the step size (0.1)
    assert rel_err(res.points, ref.points) < 1e-3
    assert (res.mask.cpu() == ref.mask).float().mean() > 0.99


def test_generic_model_route(dev):
    """Any other nn.Module runs the reference's own loop on the GPU."""
    O = _oracle()
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths

    class Torus(torch.nn.Module):
        def forward(self, x, **kw):
            q = torch.stack([x[..., :2].norm(dim=-1) - 0.6, x[..., 2]], -1)
            return O.SdfOut(sdf=q.norm(dim=-1, keepdim=True) - 0.25)

    pts = cube_cloud(2000, seed=7)
    ref = O.project_points(Torus(), pts, torch.tensor([2000]), proj_max_iters=10)
    g = pts.to(dev)
    from iso_points_amd import levelset_sampling as LS
    LS._GENERIC_WARNED.clear()
    with pytest.warns(RuntimeWarning, match="no fused SDF kernel"):       # the slow route announces itself
        res = UniformProjection()._project_points(Torus(), g, full_lengths(g), proj_max_iters=10)
    _check_projection(res, ref, Torus())


@pytest.mark.parametrize("hidden,n_layers,P", [(256, 3, 150001), (128, 2, 150001), (64, 1, 40003)])
def test_project_siren_is_repeatable_and_order_independent(dev, hidden, n_layers, P, gemm_mode):
    """Random SIREN weights make the Newton iteration chaotic: a single flipped bit anywhere shows
    up as a different end point.  The projection must be bit-identical from run to run, for a
    shuffled copy of the cloud and for a prefix of it (a point's result may not depend on which
    workgroup / lane it lands in) -- this is what makes the sharded cycle equal the single-GPU one."""
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import Siren
    torch.manual_seed(0)
    m = Siren(hidden_size=hidden, n_layers=n_layers).to(dev)
    pts = sphere_cloud(P, seed=3).to(dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8)
    ref = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    g = torch.Generator().manual_seed(7)
    for rep in range(3):
        out = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
        assert torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals), rep
        perm = torch.randperm(P, generator=g).to(dev)
        out = proj._project_points(m, pts[:, perm].contiguous(), full_lengths(pts), proj_max_iters=10)
        assert torch.equal(out.points, ref.points[:, perm]) and torch.equal(out.normals, ref.normals[:, perm]), rep
    half = pts[:, : P // 2].contiguous()
    out = proj._project_points(m, half, full_lengths(half), proj_max_iters=10)
    assert torch.equal(out.points, ref.points[:, : P // 2])
