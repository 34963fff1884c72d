"""Pin the oracle: its restatement of projection / resample must reproduce the golden vectors
that tests/golden/make_golden.py obtained by running the reference's own Python
(levelset_sampling.py with import shims) in the build container.  CPU only."""
import os

import numpy as np
import pytest
import torch

from util import assert_projection_close, rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TIGHT = 2e-6   # same float32 torch ops in the same order; slack for a different host CPU


def load(name):
    d = np.load(os.path.join(GOLD, name))
    return {k: (torch.from_numpy(d[k]) if d[k].ndim else d[k].item()) for k in d.files}


def siren_from(g):
    from oracle import iso_oracle as O
    m = O.SirenSDF(hidden_size=int(g["siren_hidden"]), n_layers=int(g["siren_layers"]))
    raw = g["siren_raw"]
    o = 0
    with torch.no_grad():
        for lin in m.lins:
            n = lin.weight.numel()
            lin.weight.copy_(raw[o:o + n].view_as(lin.weight)); o += n
            n = lin.bias.numel()
            lin.bias.copy_(raw[o:o + n]); o += n
    assert o == raw.numel()
    return m


def test_projection_sphere_cfg1():
    from oracle import iso_oracle as O
    g = load("proj_sphere_cfg1.npz")
    num = torch.tensor([g["points"].shape[1]])
    for T in (1, 10):
        r = O.project_points(O.SphereSDF(), g["points"], num, proj_max_iters=T)
        assert_projection_close(r.points, g["T%d_points" % T], tol=TIGHT)
        assert rel_err(r.normals, g["T%d_normals" % T]) < 1e-5
        assert (r.mask == g["T%d_mask" % T]).float().mean() > 0.999


def test_projection_sphere_ragged():
    from oracle import iso_oracle as O
    g = load("proj_sphere_ragged.npz")
    m = O.SphereSDF(tuple(g["center"].tolist()), float(g["radius"]))
    r = O.project_points(m, g["points"], g["num_points"], proj_max_iters=int(g["T"]))
    assert r.points.shape == g["out_points"].shape
    assert_projection_close(r.points, g["out_points"], tol=TIGHT)
    assert torch.equal(r.mask, g["out_mask"])


def test_siren_eval_and_projection():
    from oracle import iso_oracle as O
    g = load("proj_siren_small.npz")
    m = siren_from(g)
    sdf, grad = O.compute_sdf_and_grad(g["points"], m)
    assert rel_err(sdf, g["sdf"]) < TIGHT and rel_err(grad, g["grad"]) < TIGHT
    n = torch.tensor([g["points"].shape[1]])
    r = O.project_points(m, g["points"], n, proj_max_iters=int(g["T"]))
    assert rel_err(r.points, g["out_points"]) < 1e-4     # random SIREN: chaotic, 4 clamped moves
    g = load("proj_siren_fitted.npz")
    m = siren_from(g)
    n = torch.tensor([g["points"].shape[1]])
    r = O.project_points(m, g["points"], n, proj_max_iters=10)
    assert_projection_close(r.points, g["out_points"], tol=1e-5)
    r0 = O.project_points(m, g["points"], n, proj_max_iters=10, proj_tolerance=1e-30)
    assert rel_err(r0.points, g["fixed_points"]) < 1e-5


@pytest.mark.parametrize("tag", ["sphere", "sphere3", "siren"])
def test_resample(tag):
    from oracle import iso_oracle as O
    g = load("resample_%s.npz" % tag)
    m = siren_from(g) if tag == "siren" else O.SphereSDF()
    n = torch.tensor([g["points"].shape[1]])
    r = O.resample(m, g["proj_points"], g["proj_normals"], n, sample_iters=int(g["sample_iters"]),
                   knn_k=int(g["knn_k"]))
    assert_projection_close(r.points, g["out_points"], tol=1e-5)
    assert (r.mask == g["out_mask"]).float().mean() > 0.995


def test_project_points_driver_filtering():
    from oracle import iso_oracle as O
    g = load("project_points_driver.npz")
    P = g["points"].shape[1]
    r0 = O.project_points(O.SphereSDF(), g["points"], torch.tensor([P]), proj_max_iters=int(g["T"]))
    p1 = O.reduce_mask_padded(r0.points, r0.mask)
    n1 = O.reduce_mask_padded(r0.normals, r0.mask)
    r = O.resample(O.SphereSDF(), p1, n1, r0.mask.sum(-1), sample_iters=1, knn_k=int(g["knn_k"]))
    assert r.points.shape == g["levelset_points"].shape
    assert_projection_close(r.points, g["levelset_points"], tol=1e-5)


def test_splat_per_point_setup():
    """oracle per_point_info / vrk_h vs SurfaceSplatting._get_per_point_info run from the
    reference (rasterizer.py:344-563).  The reference draws a random tangent frame; any frame
    gives the same result mathematically, so the comparison is to rounding (2e-4), not bitwise."""
    from oracle import splat_oracle as SO
    g = load("splat_setup.npz")
    num = g["num"]
    pts_l = torch.split(g["points"], num.tolist())
    mx = int(num.max())
    padded = torch.zeros(len(num), mx, 3)
    for i, p in enumerate(pts_l):
        padded[i, : len(p)] = p
    h = SO.vrk_h(padded, num, float(g["frnn_radius"]))
    assert torch.equal(h, g["Vrk_h"])          # same frnn contract + same torch ops
    s = 0
    for i, n in enumerate(num.tolist()):
        M44 = g["views"][i] @ g["proj"]
        info = SO.per_point_info(g["points"][s:s + n], g["normals"][s:s + n], h[s:s + n], M44,
                                 int(g["image_size"]), float(g["cutoff"]), float(g["sigma"]))
        for k in ("radii", "ellipse_params", "cutoff_threshold", "scaler"):
            ref = g[k][s:s + n].reshape(n, -1)
            # per-point scale: b of (a,b,c) can be ~0, so normalise by the row's largest entry
            err = ((info[k].reshape(n, -1) - ref).abs() / ref.abs().amax(-1, keepdim=True)).amax(-1)
            if k == "scaler":
                # |det(Sk WJk)| is frame-invariant, but the reference's frame n x (n + rand) is
                # ill-conditioned whenever rand is nearly parallel to n: its own output then carries
                # up to ~1e-3 of noise on a fraction of a percent of the points.
                assert err.median() < 1e-6 and (err > 1e-5).float().mean() < 0.01 and err.max() < 1e-2
            else:
                assert err.max().item() < 1e-5, (k, err.max().item())
        s += n


def test_gather_with_neg_idx():
    from oracle import splat_oracle as SO
    g = load("gather_neg_idx.npz")
    assert torch.equal(SO.gather_scaler(g["scaler"], g["idx"]), g["out"])


@pytest.mark.parametrize("name", ["upsample_K16.npz", "upsample_K31.npz", "upsample_batch.npz"])
def test_upsample(name):
    """oracle upsample vs the reference's point_processing.upsample (knn_points shimmed)."""
    from oracle import iso_oracle as O
    g = load(name)
    num = g["num_points"] if "num_points" in g else None
    n_points = g["n_points"] if torch.is_tensor(g["n_points"]) else int(g["n_points"])
    up, n = O.upsample(g["points"], n_points, num_points=num, neighborhood_size=int(g["K"]))
    assert torch.equal(n, g["out_num"])
    assert up.shape == g["out_points"].shape
    assert rel_err(up, g["out_points"]) < TIGHT


def test_wlop_iterations():
    from oracle import iso_oracle as O
    g = load("wlop_ratio1.npz")
    P = g["points"]
    num = torch.tensor([P.shape[1]])
    lo, hi = P[0].min(0).values, P[0].max(0).values
    h = 4 * torch.sqrt(torch.norm(lo - hi) / P.shape[1])
    torch.manual_seed(int(g["seed"]))
    X0 = P + torch.randn_like(P[0]) * h * 0.1               # point_processing.py:59-60
    X = O.wlop_iterations(P, num, X0, num, neighborhood_size=int(g["K"]), iters=int(g["iters"]),
                          repulsion_mu=float(g["mu"]))
    assert rel_err(X, g["out_points"]) < 1e-5


def test_insert():
    from oracle import iso_oracle as O
    g = load("insert.npz")
    child, cpb = O.insert(g["ref_points"], g["ref_metrics"], g["points"], torch.tensor([g["points"].shape[1]]))
    assert torch.equal(cpb, g["child_per_batch"]) and int(cpb[0]) > 0
    assert rel_err(child, g["child_pts"]) < TIGHT


def siren_from_ref(g):
    """oracle SirenSDF carrying the weights of a golden made by the reference's own Siren class."""
    from oracle import iso_oracle as O
    m = O.SirenSDF(hidden_size=int(g["hidden"]), n_layers=int(g["n_layers"]))
    raw, o = g["raw"], 0
    with torch.no_grad():
        for lin in m.lins:
            n = lin.weight.numel()
            lin.weight.copy_(raw[o:o + n].view_as(lin.weight)); o += n
            n = lin.bias.numel()
            lin.bias.copy_(raw[o:o + n]); o += n
    assert o == raw.numel()
    return m


@pytest.mark.parametrize("name", ["siren_ref_128x2.npz", "siren_ref_256x4.npz"])
def test_siren_restatement_vs_the_reference_class(name):
    """the oracle's SirenSDF against the reference's own Siren (common.py:90-165, c_dim = 0)
    evaluated through the reference's _compute_sdf_and_grad / _project_points."""
    from oracle import iso_oracle as O
    g = load(name)
    m = siren_from_ref(g)
    sdf, grad = O.compute_sdf_and_grad(g["points"], m)
    assert rel_err(sdf, g["sdf"]) < 1e-5 and rel_err(grad, g["grad"]) < 1e-5
    r = O.project_points(m, g["points"], torch.tensor([g["points"].shape[1]]), proj_max_iters=int(g["T"]),
                         proj_tolerance=1e-30)
    assert rel_err(r.points, g["fixed_points"]) < 1e-5


def idr_from(g):
    from oracle import iso_oracle as O
    m = O.IdrSDF(hidden_size=int(g["hidden"]), n_layers=int(g["n_layers"]), skip_in=(int(g["skip"]),),
                 num_frequencies=int(g["n_freq"]))
    raw, o = g["raw"], 0
    with torch.no_grad():
        for l in range(m.num_layers - 1):
            n = m.v[l].numel()
            W = raw[o:o + n].view_as(m.v[l]); o += n
            m.v[l].copy_(W)
            m.g[l].copy_(W.norm(dim=1, keepdim=True))       # g = |v|  =>  effective weight == W
            n = m.b[l].numel()
            m.b[l].copy_(raw[o:o + n]); o += n
    assert o == raw.numel()
    return m


def test_idr_sdf_restatement():
    """oracle IdrSDF vs the reference's own SDF class (common.py:220-310) evaluated through the
    reference's _compute_sdf_and_grad / _project_points."""
    from oracle import iso_oracle as O
    g = load("idr_small.npz")
    m = idr_from(g)
    sdf, grad = O.compute_sdf_and_grad(g["points"], m)
    assert rel_err(sdf, g["sdf"]) < 1e-5 and rel_err(grad, g["grad"]) < 1e-5
    r = O.project_points(m, g["points"], torch.tensor([g["points"].shape[1]]), proj_max_iters=int(g["T"]),
                         proj_tolerance=1e-30)
    assert rel_err(r.points, g["fixed_points"]) < 1e-4


def idr_from_trace(g):
    return idr_from({"hidden": g["idr_hidden"], "n_layers": g["idr_layers"], "skip": g["idr_skip"],
                     "n_freq": g["idr_freq"], "raw": g["idr_raw"]})


def assert_trace_close(out, pts, val, mask, tol=TIGHT, stop_tol=5e-6):
    """positions / values of a sphere trace: every ray to `tol`, except rays whose |sdf| sits at the
    0.1 * 5e-5 stopping threshold (one advance of that size more or less), which must be rare.
    (`stop_tol` is raised for networks that are not 1-Lipschitz everywhere: where |grad| > 2 along
    the ray the advance p += f d amplifies rounding differences instead of damping them.)"""
    assert_projection_close(out["levelset_points"], pts, stop_tol=stop_tol, tol=tol)
    dv = (out["network_eval_on_levelset_points"].detach().cpu() - val).abs()
    assert (dv > 1e-5).float().mean() < 5e-3 and dv.max() < 4 * stop_tol
    assert (out["mask"].cpu() == mask).float().mean() > 0.995
    assert out["levelset_points_Dx"].shape == pts.shape


def test_sphere_trace_restatement():
    """oracle sphere_trace vs the reference's SphereTracing.project_points (levelset_sampling.py:679-808)."""
    from oracle import iso_oracle as O
    g = load("trace_sphere.npz")
    sph = O.SphereSDF(tuple(g["center"].tolist()), float(g["radius"]))
    out = O.sphere_trace(sph, g["ray0"], g["dirs"], proj_max_iters=10)
    assert_trace_close(out, g["T10_points"], g["T10_eval"], g["T10_mask"])
    assert 0.3 < g["T10_mask"].float().mean() < 0.95          # hits and misses are both present
    out = O.sphere_trace(sph, g["ray0"], g["dirs"], proj_max_iters=3, alpha=0.8)
    assert_trace_close(out, g["T3_points"], g["T3_eval"], g["T3_mask"])
    g = load("trace_siren.npz")
    out = O.sphere_trace(siren_from(g), g["ray0"], g["dirs"], proj_max_iters=10)
    assert_trace_close(out, g["T10_points"], g["T10_eval"], g["T10_mask"])
    g = load("trace_idr.npz")
    out = O.sphere_trace(idr_from_trace(g), g["ray0"], g["dirs"], proj_max_iters=int(g["T"]))
    assert_trace_close(out, g["out_points"], g["out_eval"], g["out_mask"])


def test_zero_crossing_restatement():
    """oracle find_zero_crossing vs the reference's find_zero_crossing_between_point_pairs + run_Secant_method
    (levelset_sampling.py:1210-1367)."""
    from oracle import iso_oracle as O
    g = load("trace_siren.npz")
    pt, mask = O.find_zero_crossing(g["ray0"], g["zc_p1"], siren_from(g), is_occupancy=False)
    assert torch.equal(mask, g["zc_mask"]) and 0.2 < mask.float().mean() < 1.0
    assert rel_err(pt, g["zc_points"]) < 1e-5
    sph = O.SphereSDF(tuple(g["center"].tolist()), float(g["radius"]))
    pt, mask = O.find_zero_crossing(g["ray0"][:, :500], g["zc_p1"][:, :500], sph, is_occupancy=False, n_steps=64,
                                    n_secant_steps=6)
    assert torch.equal(mask, g["zc_sphere_mask"])
    assert rel_err(pt, g["zc_sphere_points"]) < TIGHT


RT_CASES = {"eval": (False, {}), "train": (True, {}),
            "short_eval": (False, {"sphere_tracing_iters": 2, "n_steps": 40, "n_secant_steps": 5}),
            "short_train": (True, {"sphere_tracing_iters": 2, "n_steps": 40, "line_step_iters": 2})}
RT_CASES_SIREN = {"eval": (False, {}), "train": (True, {}),
                  "short_eval": (False, {"sphere_tracing_iters": 3, "n_steps": 64})}


def assert_raytrace_close(got, g, tag, tol=TIGHT, flip=0.0, frac=1.0):
    """points / mask / depth of RayTracing.forward.  `flip`: fraction of rays allowed a different
    network mask (a value within rounding of 0 or of the 5e-5 threshold decides the branch);
    `frac`: fraction of the mask-agreeing rays whose points and depth must agree to `tol`."""
    pts, mask, z = [t.detach().cpu() for t in got]
    same = mask == g[tag + "_mask"]
    assert (~same).float().mean() <= flip, (~same).sum()
    scale = 3.0                                              # camera distance: depths are O(3)
    dp = (pts - g[tag + "_points"]).abs().max(-1)[0] / scale
    dz = (z - g[tag + "_dist"]).abs() / scale
    ok = (dp <= tol) & (dz <= tol)
    assert ok[same].float().mean() >= frac, (tag, (~ok[same]).sum().item(), dp[same].max().item(), dz[same].max().item())


def test_ray_tracing_restatement_sphere():
    """oracle ray_tracing vs the reference's RayTracing.forward (levelset_sampling.py:831-918):
    all four branches (trace from both ends, sampler + secant, tangent-plane left-outs, minimal
    sdf samples) on an analytic sphere; every ray, bit-tight."""
    from oracle import iso_oracle as O
    g = load("raytrace_sphere.npz")
    sph = O.SphereSDF(tuple(g["center"].tolist()), float(g["radius"]))
    sdf = lambda x: sph.forward(x).sdf.reshape(-1)
    for tag, (training, kw) in RT_CASES.items():
        got = O.ray_tracing(sdf, g["cam"], g["gt"], g["dirs"], training=training,
                            uniform_steps=g[tag + "_uniform"], **kw)
        assert_raytrace_close(got, g, tag)
    m = g["eval_mask"]
    assert 0.1 < m.float().mean() < 0.6 and (g["gt"] != m).any()      # hits, misses and mask mismatches


def test_ray_tracing_restatement_siren():
    from oracle import iso_oracle as O
    g = load("raytrace_siren.npz")
    w = load("trace_siren.npz")
    net = siren_from(w)
    sdf = lambda x: net.forward(x).sdf.reshape(-1)
    for tag, (training, kw) in RT_CASES_SIREN.items():
        got = O.ray_tracing(sdf, g["cam"], g["gt"], g["dirs"], training=training,
                            uniform_steps=g[tag + "_uniform"], **kw)
        assert_raytrace_close(got, g, tag)


def test_get_tensor_values_restatement():
    """oracle get_tensor_values (grid_sample written out) vs the reference's own function
    (utils/__init__.py:325-375): bilinear / nearest / integer indexing, samples inside, on the
    border and outside [-1,1] (reflection padding)."""
    from oracle import iso_oracle as O
    g = load("image_values.npz")
    v, m = O.get_tensor_values(g["mask"], g["p"], with_mask=True, squeeze_channel_dim=True)
    assert v.shape == g["mask_bilinear"].shape and (v - g["mask_bilinear"]).abs().max() < 1e-6
    assert torch.equal(m, g["mask_valid"])
    assert (O.get_tensor_values(g["rgb"], g["p"]) - g["rgb_bilinear"]).abs().max() < 1e-6
    near = O.get_tensor_values(g["rgb"], g["p"], mode="nearest")
    assert (near != g["rgb_nearest"]).any(-1).float().mean() < 2e-3     # x.5 ties after float rounding
    assert torch.equal(O.get_tensor_values(g["sq"], g["p_in"], grid_sample=False), g["sq_index"])
    assert ((g["p"].abs() > 1).any(-1)).float().mean() > 0.2


SAMPLE_PARAMS = ((0, "weight"), (2, "bias"), (-1, "weight"), (-1, "bias"))


def sample_grads(net, value):
    net.zero_grad()
    value.backward()
    return torch.cat([getattr(net.lins[i], n).grad.reshape(-1) for i, n in SAMPLE_PARAMS]).cpu()


def test_sample_network_restatement():
    """oracle sample_network / directional_sample vs the reference's SampleNetwork /
    DirectionalSamplingNetwork (levelset_sampling.py:1170-1207, :1370-1403): sampled points and
    the gradient of a linear functional of them w.r.t. network parameters."""
    from oracle import iso_oracle as O
    g = load("sample_network.npz")
    net = siren_from(load("trace_siren.npz"))
    out, ev = O.sample_network(net, g["points"], return_eval=True)
    assert torch.equal(out.detach(), g["sn_points"]) and rel_err(ev.detach(), g["sn_eval"]) < TIGHT
    assert rel_err(sample_grads(net, (out * g["w"]).sum()), g["sn_grads"]) < 1e-5
    out, ev = O.directional_sample(net, g["points"], g["ray"], g["cam"], return_eval=True)
    assert rel_err(out.detach(), g["dn_points"]) < TIGHT and rel_err(ev.detach(), g["dn_eval"]) < TIGHT
    assert rel_err(sample_grads(net, (out * g["w"]).sum()), g["dn_grads"]) < 1e-5


EAR_CASES = {"K16": dict(knn_k=16, upsample_ratio=1.2),
             "K31_sharp": dict(knn_k=31, sharpness_angle=30, edge_sensitivity=2, upsample_ratio=1.3, repulsion_mu=0.3)}


def test_edge_aware_restatement():
    """oracle ear_tree / ear_denoise_normals / ear_upsample vs the reference's EdgeAwareProjection
    (levelset_sampling.py:442-661) on a box (a shape with edges)."""
    import math
    from oracle import iso_oracle as O
    box = O.BoxSDF()
    for tag, kw in EAR_CASES.items():
        g = load("ear_%s.npz" % tag)
        num = torch.tensor([g["points"].shape[1]])
        tree = O.ear_tree(g["points"], num, kw["knn_k"])
        sharp = 1 - math.cos(kw.get("sharpness_angle", 15) / 180 * math.pi)
        nd, wp, wn = O.ear_denoise_normals(g["points"], g["normals"], num, tree, sharp)
        assert rel_err(nd, g["denoised"]) < TIGHT and rel_err(wp, g["weights_p"]) < TIGHT
        assert rel_err(wn, g["weights_n"]) < TIGHT
        up, n = O.ear_upsample(g["points"], g["points"].shape[1], box, num.clone(), **kw)
        assert torch.equal(n, g["out_num"]) and up.shape == g["out_points"].shape
        assert rel_err(up, g["out_points"]) < TIGHT


def test_edge_aware_driver_restatement():
    """project -> resample (K-nearest tree) -> edge-aware upsample -> project, as the reference's
    EdgeAwareProjection.project_points runs it."""
    from oracle import iso_oracle as O
    g = load("ear_driver.npz")
    res = O.ear_project_points(g["points"], O.SphereSDF(), knn_k=12, sample_iters=2, upsample_ratio=1.1)
    assert res.points.shape == g["levelset_points"].shape
    assert_projection_close(res.points, g["levelset_points"], tol=TIGHT)
    assert torch.equal(res.mask, g["mask"])


@pytest.mark.parametrize("K", [16, 30])
def test_denoise_normals_restatement(K):
    """oracle denoise_normals vs the reference's point_processing.denoise_normals (:241-278)."""
    from oracle import iso_oracle as O
    g = load("denoise_normals_K%d.npz" % K)
    out = O.denoise_normals(g["points"], g["normals"], sharpness_sigma=g["sigma"], neighborhood_size=K)
    assert rel_err(out, g["out"]) < TIGHT
    assert rel_err(torch.nn.functional.normalize(g["normals"], dim=-1), g["out"]) > 0.05    # it did something


def test_ray_sampling_restatement():
    """ray_nearest_point / insurface_segments / lowest_sdf_on_segments against the
    reference's own statements (combined_modeling.py:324-386, executed by tests/golden/make_golden_rays.py)."""
    from oracle import iso_oracle as O
    g = load("ray_sampling.npz")
    B = g["cam_pos"].shape[0]
    model = O.SphereSDF(radius=float(g["sdf_radius"]))
    l0s, l1s, ps, valid_all = [], [], [], []
    for b in range(B):
        cam = g["cam_pos"][b]
        ray0 = torch.nn.functional.normalize(g["samples"][b] - cam.view(1, 3), dim=-1)
        l0, l1, valid = O.insurface_segments(cam, ray0, g["frontal%d" % b], g["occluded%d" % b])
        valid_all.append(valid)
        l0s.append(l0[valid]); l1s.append(l1[valid])
        p, _ = O.lowest_sdf_on_segments(model, cam, ray0[valid], l0[valid], l1[valid], int(g["n_points_per_ray"]))
        ps.append(p)
    assert torch.equal(torch.stack(valid_all), g["mask_insurface"].bool())
    assert rel_err(torch.cat(l0s), g["ray_len0"]) < 1e-6 and rel_err(torch.cat(l1s), g["ray_len1"]) < 1e-6
    assert rel_err(torch.cat(ps), g["p_insurface"]) < 1e-6
