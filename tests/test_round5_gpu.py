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
                outs.append(proj.project_points(pts, model, skip_upsampling=True))
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
