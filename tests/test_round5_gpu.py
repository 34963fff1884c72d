"""Round 5: the operator-API paths that issue work before a host read must give what the reference's order gives."""
import os

import pytest
import torch

from util import cube_cloud, sphere_cloud

pytestmark = pytest.mark.gpu


def test_project_points_optimistic_resample_equals_the_reference_order(dev):
    """project_points issues the resampling on the WHOLE first projection before it knows how many points converged
    (levelset_sampling.py:392-410 keeps the converged ones first).  All converged: the optimistic result is the
    result; some did not (one Newton step from a cube): it must be discarded and the reference's order taken."""
    from iso_points_amd.levelset_sampling import UniformProjection, mask_padded_to_list
    from iso_points_amd.sdf_models import SphereSDF
    model = SphereSDF().to(dev)
    for pts, iters in ((sphere_cloud(20000, seed=1).to(dev), 10), (cube_cloud(20000, seed=2).to(dev), 1)):
        outs = []
        for sync in ("1", ""):
            if sync:
                os.environ["ISO_OPAPI_SYNC"] = "1"
            else:
                os.environ.pop("ISO_OPAPI_SYNC", None)
            try:
                proj = UniformProjection(proj_max_iters=iters, knn_k=8, sample_iters=1)
                # (round 6: the optimistic order is only taken after a call that saw every point converge -- the cube case
                # is primed by hand so that the discard path stays tested)
                proj._all_converged_last = True
                outs.append(proj.project_points(pts, model, skip_upsampling=True))
                assert proj._all_converged_last == (iters == 10)
            finally:
                os.environ.pop("ISO_OPAPI_SYNC", None)
        a, b = outs
        assert a["levelset_points"].shape == b["levelset_points"].shape
        assert torch.equal(a["levelset_points"], b["levelset_points"]) and torch.equal(a["mask"], b["mask"])
        assert torch.equal(a["levelset_normals"], b["levelset_normals"])
        la, lb = mask_padded_to_list(a["levelset_points"], a["mask"]), mask_padded_to_list(b["levelset_points"], b["mask"])
        assert torch.equal(la[0], lb[0]) and torch.equal(la[0], a["levelset_points"][0][a["mask"][0]])
    assert outs[0]["levelset_points"].shape[1] < 20000          # the cube case really dropped points before resampling


def test_forward_issued_before_the_host_read_equals_the_exact_path_and_survives_an_overflow(dev):
    """SurfaceSplatting.forward rasterises before its host read once it knows a pair capacity from earlier calls; a frame
    with more point-tile pairs than 1.25 x everything seen before overflows that capacity and must take the exact path."""
    from oracle import splat_oracle as SO
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    N, S, K = 2, 128, 8
    views = torch.stack([SO.look_at_view(3.0, 20.0, 180.0 * i) for i in range(N)]).to(dev)
    projs = views @ SO.perspective(30.0).to(dev)
    rs = PointsRasterizationSettings(image_size=S, points_per_pixel=K)
    small = torch.nn.functional.normalize(sphere_cloud(3000, seed=3)[0], dim=-1).to(dev)
    big = torch.nn.functional.normalize(sphere_cloud(60000, seed=4)[0], dim=-1).to(dev)

    def run(ss, pts):
        frags, filt = ss.forward(pts, pts.clone(), cameras=(views, projs))
        return frags, filt

    ss = SurfaceSplatting(raster_settings=rs)
    f1, _ = run(ss, small)                       # exact path; learns a (small) capacity
    cap = ss._pair_cap
    f2, _ = run(ss, small)                       # early path
    assert torch.equal(f1.idx, f2.idx) and torch.equal(f1.zbuf, f2.zbuf) and torch.equal(f1.qvalue, f2.qvalue)
    f3, flt3 = run(ss, big)                      # early path overflows the small capacity -> exact path
    assert ss._pair_cap > cap
    ref, fltr = run(SurfaceSplatting(raster_settings=rs), big)
    assert torch.equal(f3.idx, ref.idx) and torch.equal(f3.zbuf, ref.zbuf) and torch.equal(f3.occupancy, ref.occupancy)
    assert torch.equal(flt3["ndc"], fltr["ndc"]) and flt3["num_points"].tolist() == fltr["num_points"].tolist()
    # and the gradient reaches the world points through the early path too
    x = small.clone().requires_grad_(True)
    fr, _ = ss.forward(x, small.clone(), cameras=(views, projs))
    (fr.occupancy.sum() + fr.zbuf[..., 0].sum()).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0


@pytest.mark.parametrize("K", [8, 5])
@pytest.mark.parametrize("case", ["pile", "wide", "pile_wide"])
def test_raster_piles_deeper_than_the_hit_list(dev, K, case):
    """k_raster's candidate-parallel path: the first 32 hits of a pixel per 256-candidate chunk go to its byte list, the
    others to the pixel's 256-bit mask; up to 32 candidates per chunk with boxes over 96 pixels go to the wide table, the
    ones beyond it are walked like the others.  Scenes that overflow each of them, bit for bit against the oracle
    (K = 8: the compile-time-K kernel; K = 5: the runtime-K one)."""
    from oracle import splat_oracle as SO
    from iso_points_amd.rasterizer import _C
    from splat_util import random_splats
    S = 48
    sc = random_splats(5000, N=2, seed=11 + K)
    if "pile" in case:
        sc["ndc"][:, :2] *= 0.08                       # everything on a patch of ~4 x 4 pixels: hundreds of hits per pixel and chunk
    if "wide" in case:
        sc["ellipse"] = (sc["ellipse"] / 36.0).contiguous()      # six times the extent: boxes of 10 - 30 pixels
        sc["radii"] = (sc["radii"] * 6.0).contiguous()
    ref = SO.splat_forward(sc["ndc"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first"], sc["num"], 0.08, S, K, bbox_or=True)
    got = _C.splat_points(sc["ndc"].to(dev), sc["ellipse"].to(dev), sc["cutoff"].to(dev), sc["radii"].to(dev),
                          sc["first"].to(dev), sc["num"].to(dev), 0.08, S, K, 0, 0)
    for g, r, nm in zip(got, ref, ("idx", "zbuf", "qvalue", "occupancy")):
        assert torch.equal(g.cpu(), r), "%s differs (%d entries)" % (nm, (g.cpu() != r).sum().item())
    hit = (ref[0][..., 0] >= 0).float().mean().item()
    assert hit > (0.002 if case == "pile" else 0.05), hit           # (the scene does land on the image)


def test_drawn_tiles_equal_static_shares(dev):
    """The step kernels' workgroups draw their next tile from a counter (siren_x3.hip / idr_x16.hip); a point's result
    does not depend on the tile it sits in or on the workgroup that takes it: projections and evaluations are
    bit-identical with every gridDim-th tile (iso_*_set_drawn_tiles(0)), for sizes around the tile shapes' rounds."""
    from iso_points_amd import _lib
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import Siren, siren_sdf_and_grad, idr_sdf_and_grad
    from oracle import iso_oracle as O
    lib = _lib.load()
    torch.manual_seed(3)
    siren = Siren(hidden_size=256, n_layers=3).to(dev)
    idr = O.IdrSDF(hidden_size=256, n_layers=4, skip_in=(2,), num_frequencies=4).to(dev)
    for prm in list(siren.parameters()) + list(idr.parameters()):
        prm.requires_grad_(False)
    proj = UniformProjection()
    for P in (20000, 24576 + 96, 70001):
        pts = (sphere_cloud(P, seed=P) * 0.9).to(dev)
        outs = []
        for on in (1, 0):
            assert lib.iso_siren_set_drawn_tiles(on) == 0 and lib.iso_idr_set_drawn_tiles(on) == 0
            try:
                r = proj._project_points(siren, pts, full_lengths(pts), proj_max_iters=6)
                s, g = siren_sdf_and_grad(siren, pts[0])
                ri = proj._project_points(idr, pts, full_lengths(pts), proj_max_iters=3)
                si, gi = idr_sdf_and_grad(idr, pts[0])
                torch.cuda.synchronize()
            finally:
                lib.iso_siren_set_drawn_tiles(-1); lib.iso_idr_set_drawn_tiles(-1)
            outs.append((r.points, r.normals, r.mask, s, g, ri.points, ri.mask, si, gi))
        for a, b in zip(*outs):
            assert torch.equal(a, b), P
