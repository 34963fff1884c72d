"""Ray queries on the fused SDF kernels: SphereTracing.project_points
(iso_trace_{sphere,siren,idr}), the value-only evaluation, and find_zero_crossing_between_point_pairs /
run_Secant_method on top of it -- against the golden vectors made by the reference's own functions
(tests/golden/make_golden_trace.py) and against the oracle on seeded inputs."""
import pytest
import torch

from test_oracle_golden import assert_trace_close, idr_from_trace, load, siren_from
from util import assert_projection_close, cube_cloud, fitted_siren, rel_err

pytestmark = pytest.mark.gpu


def _O():
    from oracle import iso_oracle as O
    return O


def _sphere(dev, g=None, center=(0.0, 0.0, 0.0), radius=1.0):
    from iso_points_amd.sdf_models import SphereSDF
    if g is not None:
        center, radius = tuple(g["center"].tolist()), float(g["radius"])
    return SphereSDF(center, radius).to(dev)


def test_trace_sphere_golden(dev):
    from iso_points_amd.levelset_sampling import SphereTracing
    g = load("trace_sphere.npz")
    m = _sphere(dev, g)
    out = SphereTracing(proj_max_iters=10).project_points(g["ray0"].to(dev), g["dirs"].to(dev), m)
    assert_trace_close(out, g["T10_points"], g["T10_eval"], g["T10_mask"])
    out = SphereTracing(proj_max_iters=3, alpha=0.8).project_points(g["ray0"].to(dev), g["dirs"].to(dev), m)
    assert_trace_close(out, g["T3_points"], g["T3_eval"], g["T3_mask"])
    assert out["levelset_points"].shape == g["ray0"].shape and out["mask"].dtype == torch.bool


@pytest.mark.parametrize("mode", ["split16", "f32"])
def test_trace_siren_golden(dev, mode):
    from iso_points_amd import _lib
    from iso_points_amd.levelset_sampling import SphereTracing
    g = load("trace_siren.npz")
    m = siren_from(g).to(dev)
    old = _lib.load().iso_siren_get_gemm_mode()
    _lib.call("iso_siren_set_gemm_mode", 1 if mode == "split16" else 0)
    try:
        out = SphereTracing(proj_max_iters=10).project_points(g["ray0"].to(dev), g["dirs"].to(dev), m)
    finally:
        _lib.call("iso_siren_set_gemm_mode", old)
    assert_trace_close(out, g["T10_points"], g["T10_eval"], g["T10_mask"], tol=1e-5)


def test_trace_idr_golden(dev):
    from iso_points_amd.levelset_sampling import SphereTracing
    g = load("trace_idr.npz")
    m = idr_from_trace(g).to(dev)
    out = SphereTracing(proj_max_iters=int(g["T"])).project_points(g["ray0"].to(dev), g["dirs"].to(dev), m)
    assert_trace_close(out, g["out_points"], g["out_eval"], g["out_mask"], tol=1e-5)


def _rays(n, seed, dev, start_radius=1.05):
    g = torch.Generator().manual_seed(seed)
    r0 = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * start_radius
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.6
    d = torch.nn.functional.normalize(tgt - r0, dim=-1)
    return r0, d


@pytest.mark.parametrize("H,L", [(256, 3), (128, 2), (64, 2)])
def test_trace_siren_vs_oracle(dev, H, L):
    """fitted SIREN (a real SDF: rays converge) -- every kernel family (x3 H=256/128, f32 H=64)."""
    O = _O()
    from iso_points_amd.levelset_sampling import SphereTracing
    m = fitted_siren(O, H, L, seed=1, fit=150)
    r0, d = _rays(4000, 21, dev)
    ref = O.sphere_trace(m, r0, d, proj_max_iters=12, alpha=0.9)
    out = SphereTracing(proj_max_iters=12, alpha=0.9).project_points(r0.to(dev), d.to(dev), m.to(dev))
    # A briefly fitted SIREN is an SDF near the sphere only: rays that miss wander through regions
    # where |grad| > 2 and the advance p += f d amplifies rounding differences, so positions are
    # compared on the rays that hit (and the hit/miss decision on all of them).
    hit = ref["mask"]
    assert 0.3 < hit.float().mean() < 0.99
    assert (out["mask"].cpu() == hit).float().mean() > 0.995
    assert_projection_close(out["levelset_points"].cpu()[hit], ref["levelset_points"][hit], stop_tol=1e-4, tol=1e-5)
    miss = ~hit
    assert rel_err(out["levelset_points"].cpu()[miss], ref["levelset_points"][miss]) < 0.3   # same wandering, roughly


@pytest.mark.parametrize("H,NL,skip,NF", [(512, 8, (4,), 6), (128, 3, (1,), 0), (256, 4, (3,), 10)])
def test_trace_idr_vs_oracle(dev, H, NL, skip, NF):
    """geometric-init IDR networks (~ sphere of radius 0.6): feature-split, staged (H=128) and
    wide-encoding (F=10 -> staged) kernels."""
    O = _O()
    from iso_points_amd.levelset_sampling import SphereTracing
    torch.manual_seed(H + NL)
    m = O.IdrSDF(hidden_size=H, n_layers=NL, skip_in=skip, num_frequencies=NF)
    r0, d = _rays(1500, 22, dev)
    ref = O.sphere_trace(m, r0, d, proj_max_iters=8)
    out = SphereTracing(proj_max_iters=8).project_points(r0.to(dev), d.to(dev), m.to(dev))
    assert_trace_close(out, ref["levelset_points"], ref["network_eval_on_levelset_points"], ref["mask"], tol=2e-5)


def test_value_only_evaluation_equals_the_full_one(dev):
    """grad_out = NULL skips the reverse sweep; the value must not change by a single bit."""
    O = _O()
    from iso_points_amd.sdf_models import FusedSdf, idr_sdf_and_grad, siren_sdf_and_grad
    pts = cube_cloud(5000, seed=3)[0].to(dev)
    for H, L in ((256, 3), (128, 1), (64, 2)):
        torch.manual_seed(H)
        m = O.SirenSDF(hidden_size=H, n_layers=L).to(dev)
        full, grad = siren_sdf_and_grad(m, pts)
        val, none = siren_sdf_and_grad(m, pts, need_grad=False)
        assert none is None and torch.equal(full, val)
        assert torch.equal(FusedSdf(m, dev)(pts.view(50, 100, 3)), full.view(50, 100))
    for H, NL, skip, NF in ((512, 8, (4,), 6), (128, 3, (1,), 0), (256, 5, (), 4)):
        torch.manual_seed(H)
        m = O.IdrSDF(hidden_size=H, n_layers=NL, skip_in=skip, num_frequencies=NF).to(dev)
        full, grad = idr_sdf_and_grad(m, pts)
        val, none = idr_sdf_and_grad(m, pts, need_grad=False)
        assert none is None
        # H = 128 evaluates the value with the feature-split kernel and the gradient with the staged one
        assert torch.equal(full, val) if H != 128 else rel_err(val, full) < 1e-6
    sph = _sphere(dev, center=(0.1, 0.0, 0.0), radius=0.5)
    assert torch.equal(FusedSdf(sph, dev)(pts), sph(pts).sdf.view(-1))


def test_trace_edge_cases(dev):
    from iso_points_amd.levelset_sampling import SphereTracing
    O = _O()
    m = _sphere(dev, radius=0.5)
    st = SphereTracing(proj_max_iters=10)
    # no rays
    out = st.project_points(torch.zeros(1, 0, 3, device=dev), torch.zeros(1, 0, 3, device=dev), m)
    assert out["levelset_points"].shape == (1, 0, 3) and out["mask"].shape == (1, 0)
    # rays that miss: they run out of the bounding sphere, stay at their last inside position, mask False
    r0 = torch.tensor([[0.0, 0.9, -0.5], [0.0, 0.0, -1.0]], device=dev)
    d = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], device=dev)
    out = SphereTracing(proj_max_iters=40).project_points(r0, d, m)
    ref = O.sphere_trace(O.SphereSDF(radius=0.5), r0.cpu(), d.cpu(), proj_max_iters=40)
    assert out["mask"].tolist() == [False, True] == ref["mask"].tolist()
    assert rel_err(out["levelset_points"], ref["levelset_points"]) < 1e-6
    assert float(out["levelset_points"][0].norm()) < 1.1
    assert abs(float(out["levelset_points"][1, 2]) + 0.5) < 5e-5
    # zero iterations: one evaluation, no advance
    out = SphereTracing(proj_max_iters=0).project_points(r0, d, m)
    assert torch.equal(out["levelset_points"], r0)
    assert rel_err(out["network_eval_on_levelset_points"], m(r0).sdf.view(-1)) < 1e-6
    # CPU tensors are rejected: there is no CPU path
    with pytest.raises(RuntimeError):
        st.project_points(r0.cpu(), d.cpu(), m)
    # a model without a fused kernel takes the generic route (the reference's loop on the GPU)

    class Torus(torch.nn.Module):
        def forward(self, x, **kw):
            from iso_points_amd.sdf_models import NetOutput
            q = torch.stack([x[..., [0, 2]].norm(dim=-1) - 0.6, x[..., 1]], -1)
            return NetOutput(sdf=q.norm(dim=-1, keepdim=True) - 0.2)

    r0, d = _rays(500, 5, dev)
    out = st.project_points(r0.to(dev), d.to(dev), Torus().to(dev))
    ref = O.sphere_trace(Torus(), r0, d, proj_max_iters=10)
    assert_trace_close(out, ref["levelset_points"], ref["network_eval_on_levelset_points"], ref["mask"], tol=1e-5)


def test_zero_crossing_golden(dev):
    from iso_points_amd.levelset_sampling import find_zero_crossing_between_point_pairs
    g = load("trace_siren.npz")
    m = siren_from(g).to(dev)
    pt, mask = find_zero_crossing_between_point_pairs(g["ray0"].to(dev), g["zc_p1"].to(dev), m, is_occupancy=False)
    same = (mask.cpu() == g["zc_mask"])
    assert same.float().mean() > 0.998            # a proposal value within rounding of 0 may flip a sign
    both = (mask.cpu() & g["zc_mask"])
    assert rel_err(pt.cpu()[both], g["zc_points"][both]) < 1e-5
    assert bool((pt.cpu()[~mask.cpu()] == 1).all())
    sph = _sphere(dev, g)
    pt, mask = find_zero_crossing_between_point_pairs(g["ray0"][:, :500].to(dev), g["zc_p1"][:, :500].to(dev), sph,
                                                      is_occupancy=False, n_steps=64, n_secant_steps=6)
    assert torch.equal(mask.cpu(), g["zc_sphere_mask"])
    assert rel_err(pt, g["zc_sphere_points"]) < 1e-5


def test_sample_networks_golden(dev):
    """SampleNetwork / DirectionalSamplingNetwork (Eq. 13) with D_xF from the fused kernel: sampled
    points and parameter gradients vs the reference's own layers (tests/golden/make_golden_sample.py)."""
    from test_oracle_golden import sample_grads
    from iso_points_amd.levelset_sampling import DirectionalSamplingNetwork, SampleNetwork
    g = load("sample_network.npz")
    net = siren_from(load("trace_siren.npz")).to(dev)
    w = g["w"].to(dev)
    out, ev = SampleNetwork()(net, g["points"].to(dev), return_eval=True)
    assert torch.equal(out.detach().cpu(), g["sn_points"]) and rel_err(ev.detach().cpu(), g["sn_eval"]) < 1e-5
    assert rel_err(sample_grads(net, (out * w).sum()), g["sn_grads"]) < 2e-5
    out, ev = DirectionalSamplingNetwork()(net, g["points"].to(dev), g["ray"].to(dev), g["cam"].to(dev), return_eval=True)
    assert rel_err(out.detach().cpu(), g["dn_points"]) < 2e-6 and rel_err(ev.detach().cpu(), g["dn_eval"]) < 1e-5
    # the parameter gradient is a cancelling sum over points of 1 / (D_xF . v) terms: any two f32
    # evaluations of it (the reference on the CPU, torch autograd on this GPU, the fused kernel)
    # differ by a few 1e-5.  Judge against the float64 value of the same statement: no farther
    # from it than twice the reference's own f32 result is.
    got = sample_grads(net, (out * w).sum())
    assert rel_err(got, g["dn_grads"]) < 1e-4
    import copy
    net64 = copy.deepcopy(net).cpu().double()
    o64 = _O().directional_sample(net64, g["points"].double(), g["ray"].double(), g["cam"].double())
    truth = sample_grads(net64, (o64 * g["w"].double()).sum())
    err_ref = rel_err(g["dn_grads"].double(), truth)
    assert rel_err(got.double(), truth) < max(2 * err_ref, 2e-5), (rel_err(got.double(), truth), err_ref)
    # a module without a fused kernel goes through autograd
    sph = _sphere(dev, center=(0.0, 0.0, 0.0), radius=0.7)
    out = SampleNetwork()(sph, g["points"].to(dev))
    assert torch.equal(out.cpu(), g["points"])
    with pytest.raises(RuntimeError):
        SampleNetwork()(net, g["points"])
