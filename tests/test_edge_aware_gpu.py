"""EdgeAwareProjection (levelset_sampling.py:442-661) on the library's neighbour search, gathers
and fused SDF gradient: against the golden vectors made by the reference's own class
(tests/golden/make_golden_ear.py) and against the oracle on a SIREN."""
import pytest
import torch

from test_oracle_golden import EAR_CASES, load
from util import fitted_siren, rel_err

pytestmark = pytest.mark.gpu


def _O():
    from oracle import iso_oracle as O
    return O


@pytest.mark.parametrize("tag", list(EAR_CASES))
def test_edge_aware_golden(dev, tag):
    from iso_points_amd.levelset_sampling import EdgeAwareProjection
    g = load("ear_%s.npz" % tag)
    box = _O().BoxSDF().to(dev)
    pts = g["points"].to(dev)
    num = torch.tensor([pts.shape[1]], device=dev)
    ear = EdgeAwareProjection(**EAR_CASES[tag])
    idx = ear._create_tree(pts, refresh_tree=True, num_points_per_cloud=num)
    assert idx.shape == (1, pts.shape[1], EAR_CASES[tag]["knn_k"]) and idx.dtype == torch.int64
    assert ear._create_tree(pts, refresh_tree=False) is idx
    nd, wp, wn = ear.denoise_normals(pts, g["normals"].to(dev), num)
    assert rel_err(nd, g["denoised"]) < 1e-5 and rel_err(wp, g["weights_p"]) < 1e-5 and rel_err(wn, g["weights_n"]) < 1e-5
    up, n = ear.upsample(pts, pts.shape[1], box, num.clone())
    assert torch.equal(n.cpu(), g["out_num"]) and up.shape == g["out_points"].shape
    P = pts.shape[1]
    assert rel_err(up[:, -P:], g["out_points"][:, -P:]) < 1e-5          # the LOP-moved input points
    # the inserted points are the candidates of the max_P sparsest fathers of each round, and every
    # round's neighbourhoods depend on the previous one's insertions: a rounding-level difference
    # (here: which axis the box's inner gradient picks for an off-surface point) swaps a row and
    # the swap propagates -- the goldens run 2-3 rounds and a few rows may differ
    d = (up[0, :-P].cpu() - g["out_points"][0, :-P]).abs().amax(-1)
    assert (d > 1e-5).float().mean() < 0.03, int((d > 1e-5).sum())


def test_edge_aware_siren_vs_oracle(dev):
    """the same through the fused SIREN gradient kernel, then the inherited projection"""
    O = _O()
    from iso_points_amd.levelset_sampling import EdgeAwareProjection
    net = fitted_siren(O, 128, 2, seed=2, fit=150)
    gen = torch.Generator().manual_seed(6)
    p = torch.nn.functional.normalize(torch.randn(1, 1500, 3, generator=gen), dim=-1)
    num = torch.tensor([1500])
    pts = O.project_points(net, p, num, proj_max_iters=10).points
    ref, n_ref = O.ear_upsample(pts, 1500, net, num.clone(), knn_k=12, upsample_ratio=1.2)
    ear = EdgeAwareProjection(knn_k=12, upsample_ratio=1.2)
    up, n = ear.upsample(pts.to(dev), 1500, net.to(dev), num.to(dev))
    assert torch.equal(n.cpu(), n_ref) and int(n) == 1800
    # an inserted point is the candidate of one of the max_P sparsest fathers: a rounding-level
    # difference in the network gradient can swap two fathers at the cut -- allow a few rows
    d = (up.cpu() - ref).abs().amax(-1)
    assert (d > 1e-5).float().mean() < 0.01, (d > 1e-5).sum()
    res = ear._project_points(net, up, n, proj_max_iters=10)
    assert res.mask.float().mean() > 0.99
    with pytest.raises(NotImplementedError):
        ear.upsample(torch.cat([pts, pts]).to(dev), 1500, net, None)


@pytest.mark.parametrize("sens", [1, 2, 1.5])
def test_ear_candidates_kernel(dev, sens):
    """The fused K^2 scan == the reference's (N,P,K,K,3) tensor expression (:609-628)."""
    from iso_points_amd import _lib
    gen = torch.Generator().manual_seed(4)
    P, K = 3000, 31
    pts = torch.rand(1, P, 3, generator=gen)
    nrm = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=gen), dim=-1)
    knn = pts[:, :, None, :] + 0.05 * torch.randn(1, P, K, 3, generator=gen)
    knn_n = torch.nn.functional.normalize(nrm[:, :, None, :] + 0.5 * torch.randn(1, P, K, 3, generator=gen), dim=-1)
    mid = (knn + 2 * pts[..., None, :]) / 3
    d = mid.unsqueeze(-2) - knn.unsqueeze(-3)
    edge = (2 - torch.sum(nrm.unsqueeze(-2) * knn_n, dim=-1)) ** sens
    m = torch.norm(d, dim=-1) - torch.sum((d * knn_n.unsqueeze(-2)) ** 2, dim=-1)
    m = m.min(dim=-1)[0].abs().clamp_min(1e-17).sqrt()
    sp_ref, nb = (edge * m).max(dim=-1)
    cand_ref = torch.gather(mid, 2, nb[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
    sp = torch.empty(1, P, device=dev)
    cand = torch.empty(1, P, 3, device=dev)
    a = [t.to(dev).contiguous() for t in (pts, nrm, knn, knn_n)]
    _lib.call("iso_ear_candidates", *[_lib.ptr(t) for t in a], P, K, float(sens), _lib.ptr(sp), _lib.ptr(cand),
              _lib.stream())
    assert rel_err(sp, sp_ref) < 2e-6
    assert ((cand.cpu() - cand_ref).abs().amax(-1) > 1e-6).float().mean() < 2e-3     # near-equal maxima


def test_edge_aware_driver_golden(dev):
    """EdgeAwareProjection.project_points (inherited driver: project -> resample on the K-nearest
    tree -> edge-aware upsample -> project) vs the reference's own run."""
    from iso_points_amd.levelset_sampling import EdgeAwareProjection
    from iso_points_amd.sdf_models import SphereSDF
    g = load("ear_driver.npz")
    ear = EdgeAwareProjection(knn_k=12, sample_iters=2, upsample_ratio=1.1)
    out = ear.project_points(g["points"].to(dev), SphereSDF().to(dev))
    assert out["levelset_points"].shape == g["levelset_points"].shape
    assert torch.equal(out["mask"].cpu(), g["mask"])
    P = g["points"].shape[1]
    d = (out["levelset_points"][0].cpu() - g["levelset_points"][0]).abs().amax(-1)
    assert (d[-P:] > 1e-5).float().mean() < 5e-3            # the resampled input points
    assert (d[:-P] > 1e-5).float().mean() < 0.03            # the inserted ones (a cut in a sorted list)
