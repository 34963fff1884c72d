"""upsample / insert / FPS / wlop on the GPU vs the golden vectors of the reference's own
functions and vs the oracle."""
import pytest
import torch

from test_oracle_golden import load
from util import rel_err, sphere_cloud

pytestmark = pytest.mark.gpu


def _O():
    from oracle import iso_oracle as O
    return O


@pytest.mark.parametrize("K", [1, 9, 32])
def test_knn_points(dev, K):
    O = _O()
    from iso_points_amd.point_processing import knn_points
    g = torch.Generator().manual_seed(K)
    p1 = torch.rand(2, 300, 3, generator=g)
    p2 = torch.rand(2, 500, 3, generator=g)
    l1, l2 = torch.tensor([300, 120]), torch.tensor([500, 20])
    ref = O.knn_points(p1, p2, l1, l2, K=K)
    got = knn_points(p1.to(dev), p2.to(dev), l1.to(dev), l2.to(dev), K=K, return_nn=True)
    for b in range(2):
        n = int(l1[b])
        assert torch.equal(got.idx[b, :n].cpu(), ref.idx[b, :n])
        assert torch.equal(got.dists[b, :n].cpu(), ref.dists[b, :n])
        assert torch.equal(got.knn[b, :n].cpu(), ref.knn[b, :n])


@pytest.mark.parametrize("name", ["upsample_K16.npz", "upsample_K31.npz", "upsample_batch.npz"])
def test_upsample_golden(dev, name):
    from iso_points_amd.point_processing import upsample
    g = load(name)
    num = g["num_points"].to(dev) if "num_points" in g else None
    n_points = g["n_points"].to(dev) if torch.is_tensor(g["n_points"]) else int(g["n_points"])
    up, n = upsample(g["points"].to(dev), n_points, num_points=num, neighborhood_size=int(g["K"]))
    assert torch.equal(n.cpu(), g["out_num"])
    assert up.shape == g["out_points"].shape
    assert rel_err(up, g["out_points"]) < 1e-5


def test_upsample_candidates_kernel(dev):
    """The fused K^2 scan == the reference's (N,P,K,K) tensor expression."""
    from iso_points_amd.point_processing import _upsample_candidates
    g = torch.Generator().manual_seed(3)
    P, K = 2000, 31
    pts = torch.rand(1, P, 3, generator=g)
    knn = pts[:, :, None, :] + 0.05 * torch.randn(1, P, K, 3, generator=g)
    mid = (knn + 2 * pts[..., None, :]) / 3
    md = torch.norm(mid.unsqueeze(-2) - knn.unsqueeze(-3), dim=-1).min(dim=-1)[0]
    sp_ref, nb = md.max(dim=-1)
    cand_ref = torch.gather(mid, 2, nb[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
    sp, cand = _upsample_candidates(pts.to(dev), knn.to(dev))
    assert rel_err(sp, sp_ref) < 1e-6
    assert rel_err(cand, cand_ref) < 1e-6


def test_fps(dev):
    O = _O()
    from iso_points_amd.point_processing import farthest_sampling
    from iso_points_amd.levelset_sampling import with_host_lengths
    p = sphere_cloud(3000, seed=9)
    pb = torch.cat([p, sphere_cloud(3000, seed=10)], dim=0)
    lens = [3000, 1777]
    num = with_host_lengths(torch.tensor(lens, device=dev), lens)
    smp, ns, idx = farthest_sampling(pb.to(dev), num, 0.25)
    for b in range(2):
        n = int(ns[b])
        ref = O.farthest_point_sampling(pb[b, :lens[b]], n, start=0)
        assert torch.equal(idx[b, :n].cpu(), ref)
        assert torch.equal(smp[b, :n].cpu(), pb[b][ref])
    # coverage property: FPS radius shrinks monotonically
    d = torch.cdist(pb[0, :3000], smp[0, :int(ns[0])].cpu()).min(dim=1).values.max()
    assert d < 0.12


def test_fps_small_clouds_in_registers(dev, monkeypatch):
    """clouds below 8 k points take k_fps_reg (points and min-distances in registers, no memory access per sample): the
    memory-walking kernel's sequence on a ragged batch, a lattice (massive distance ties), duplicates sampled past
    exhaustion, and a non-zero start."""
    from iso_points_amd.point_processing import farthest_sampling
    from iso_points_amd.levelset_sampling import with_host_lengths
    lat = torch.stack(torch.meshgrid(*([torch.arange(14.0)] * 3), indexing="ij"), -1).view(1, -1, 3) * 0.1
    dup = sphere_cloud(3000, seed=5); dup[0, 1500:] = dup[0, :1500]
    rag = torch.cat([sphere_cloud(8000, seed=6), sphere_cloud(8000, seed=7)], dim=0)
    for pts, lens, ratio in ((lat, [2744], 1.0), (dup, [3000], 0.9), (rag, [8000, 4321], 0.5), (sphere_cloud(700, seed=8), [700], 0.3)):
        num = with_host_lengths(torch.tensor(lens, device=dev), lens)
        g = torch.Generator().manual_seed(11)
        a = farthest_sampling(pts.to(dev), num, ratio, random_start=True, generator=g)
        monkeypatch.setenv("ISO_FPS_ONE_WORKGROUP", "1")
        g = torch.Generator().manual_seed(11)
        b = farthest_sampling(pts.to(dev), num, ratio, random_start=True, generator=g)
        monkeypatch.delenv("ISO_FPS_ONE_WORKGROUP")
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]), lens


def test_fps_grid_wide_form(dev, monkeypatch):
    """clouds of >= 8 k points take the cooperative grid-wide kernel: same sample sequence as the
    oracle and as the one-workgroup kernel, ragged batch, duplicated points (ties -> lowest index)."""
    O = _O()
    from iso_points_amd.point_processing import farthest_sampling
    from iso_points_amd.levelset_sampling import with_host_lengths
    pb = torch.cat([sphere_cloud(20000, seed=21), sphere_cloud(20000, seed=22)], dim=0)
    pb[1, 5000:10000] = pb[1, :5000]                       # exact duplicates: equal distances
    lens = [20000, 12345]
    num = with_host_lengths(torch.tensor(lens, device=dev), lens)
    smp, ns, idx = farthest_sampling(pb.to(dev), num, 0.02)
    for b in range(2):
        n = int(ns[b])
        ref = O.farthest_point_sampling(pb[b, :lens[b]], n, start=0)
        assert torch.equal(idx[b, :n].cpu(), ref)
    monkeypatch.setenv("ISO_FPS_ONE_WORKGROUP", "1")
    smp1, ns1, idx1 = farthest_sampling(pb.to(dev), num, 0.02)
    monkeypatch.delenv("ISO_FPS_ONE_WORKGROUP")
    assert torch.equal(idx, idx1) and torch.equal(smp, smp1)
    # a large cloud (several workgroups, 8 points per thread): grid form == one-workgroup form
    big = torch.nn.functional.normalize(sphere_cloud(300000, seed=23), dim=-1).to(dev)
    nb = torch.tensor([300000], device=dev)
    a = farthest_sampling(big, nb, 0.002)[2]
    monkeypatch.setenv("ISO_FPS_ONE_WORKGROUP", "1")
    b = farthest_sampling(big, nb, 0.002)[2]
    monkeypatch.delenv("ISO_FPS_ONE_WORKGROUP")
    assert torch.equal(a, b) and a.shape[1] == 600 and len(set(a[0].tolist())) == 600


def test_wlop_golden_and_subsample(dev):
    O = _O()
    from iso_points_amd.point_processing import wlop
    g = load("wlop_ratio1.npz")
    P = g["points"]
    gen = torch.Generator().manual_seed(int(g["seed"]))
    X, nX = wlop(P.to(dev), None, ratio=1.0, neighborhood_size=int(g["K"]), iters=int(g["iters"]),
                 repulsion_mu=float(g["mu"]), generator=gen)
    # same CPU randn stream as the reference run (torch.manual_seed + randn_like)
    assert rel_err(X, g["out_points"]) < 1e-5
    # ratio < 1: FPS start + the same LOP iterations as the oracle
    P2 = sphere_cloud(4000, seed=33, jitter=0.02)
    X2, n2 = wlop(P2.to(dev), None, ratio=0.25, perturb=False)
    sel = O.farthest_point_sampling(P2[0], 1000, start=0)
    ref = O.wlop_iterations(P2, torch.tensor([4000]), P2[:, sel], torch.tensor([1000]))
    assert int(n2[0]) == 1000
    assert rel_err(X2, ref) < 1e-5


def test_insert_golden(dev):
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    g = load("insert.npz")

    class Ref(object):
        def __init__(s):
            s.p, s.f = g["ref_points"].to(dev), g["ref_metrics"].to(dev)
        def points_packed(s): return s.p
        def features_packed(s): return s.f
        def num_points_per_cloud(s): return torch.tensor([s.p.shape[0]], device=dev)
        def __len__(s): return 1

    pts = g["points"].to(dev)
    _, num_after, child, cpb = UniformProjection(knn_k=8).insert(Ref(), pts, full_lengths(pts))
    assert torch.equal(cpb.cpu(), g["child_per_batch"])
    assert rel_err(child, g["child_pts"]) < 1e-6
    assert int(num_after[0]) == pts.shape[1] + int(cpb[0])


def test_project_points_with_upsampling_and_insert(dev):
    """UniformProjection.project_points full driver: project -> resample -> upsample -> project
    (levelset_sampling.py:353-439) and the ref_pcl branch."""
    from iso_points_amd.levelset_sampling import UniformProjection
    from iso_points_amd.sdf_models import SphereSDF
    g = torch.Generator().manual_seed(5)
    pts = ((torch.rand(1, 3000, 3, generator=g) - 0.5) * 2.6).to(dev)
    proj = UniformProjection(proj_max_iters=3, knn_k=8)
    out = proj.project_points(pts, SphereSDF().to(dev))
    assert out["levelset_points"].shape[1] == 3000          # upsample refills to the input count
    assert out["mask"].float().mean() > 0.95
    r = out["levelset_points"][out["mask"]].norm(dim=-1)
    assert (r - 1).abs().max() < 1e-3

    class Ref(object):
        def __init__(s):
            s.p = sphere_cloud(500, seed=7, jitter=0.0)[0].to(dev)
            s.f = torch.exp(3 * torch.randn(500, 1, generator=torch.Generator().manual_seed(8))).to(dev)
        def points_packed(s): return s.p
        def features_packed(s): return s.f
        def num_points_per_cloud(s): return torch.tensor([500], device=dev)
        def __len__(s): return 1
    out2 = proj.project_points(sphere_cloud(3000, seed=6).to(dev), SphereSDF().to(dev), ref_pcl=Ref(),
                               proj_max_iters=10)
    assert out2["levelset_points"].shape[1] > 3000


def test_sample_uniform_iso_points(dev):
    """levelset_sampling.py:1405-1445 end to end: n uniformly spread iso-points of a sphere."""
    from iso_points_amd.levelset_sampling import sample_uniform_iso_points
    from iso_points_amd.sdf_models import SphereSDF
    gen = torch.Generator().manual_seed(0)
    n = 2000
    pts = sample_uniform_iso_points(SphereSDF(radius=0.8).to(dev), n, bounding_sphere_radius=1.0, generator=gen,
                                    device=dev)
    assert pts.shape[0] == 1 and abs(pts.shape[1] - n) <= n // 50
    assert ((pts[0].norm(dim=-1) - 0.8).abs() < 1e-3).all()
    # uniformity: nearest-neighbour spacing is tight around its mean (random samples have cv ~ 0.5)
    d = torch.cdist(pts[0], pts[0])
    d.fill_diagonal_(10.0)
    nn = d.min(dim=1).values
    assert (nn.std() / nn.mean()).item() < 0.35


@pytest.mark.parametrize("K", [16, 30])
def test_denoise_normals_golden(dev, K):
    from iso_points_amd.point_processing import denoise_normals
    g = load("denoise_normals_K%d.npz" % K)
    out = denoise_normals(g["points"].to(dev), g["normals"].to(dev), sharpness_sigma=g["sigma"], neighborhood_size=K)
    assert out.shape == g["out"].shape and rel_err(out, g["out"]) < 1e-5
    with pytest.raises(NotImplementedError):
        denoise_normals(torch.cat([g["points"], g["points"]]).to(dev), torch.cat([g["normals"], g["normals"]]).to(dev))


def test_insert_batch_of_ragged_clouds(dev):
    """insert() on a batch: cloud 0 = the golden cloud, cloud 1 = its first 700 points (padded): cloud 0's children are
    the golden's, each cloud's children follow from its own valid points only (levelset_sampling.py:172-233)."""
    from iso_points_amd.levelset_sampling import UniformProjection, with_host_lengths
    g = load("insert.npz")

    class Ref(object):
        def __init__(s):
            s.p, s.f = g["ref_points"].to(dev), g["ref_metrics"].to(dev)
        def points_packed(s): return s.p
        def features_packed(s): return s.f
        def num_points_per_cloud(s): return torch.tensor([s.p.shape[0]], device=dev)
        def __len__(s): return 1

    one = g["points"].to(dev)                                      # (1,P,3)
    P = one.shape[1]
    two = torch.cat([one, one.clone()], dim=0)
    two[1, 700:] = 0
    lens = with_host_lengths(torch.tensor([P, 700], dtype=torch.int64, device=dev), [P, 700])
    proj = UniformProjection(knn_k=8)
    grown, n_after, child, cpb = proj.insert(Ref(), two, lens)
    assert int(cpb[0]) == int(g["child_per_batch"][0]) and rel_err(child[0, :int(cpb[0])], g["child_pts"][0]) < 1e-6
    # cloud 1 alone (same bounding box: the padding rows are zeros inside it)
    solo = two[1:2].clone()
    _, _, child1, cpb1 = proj.insert(Ref(), torch.cat([solo, solo]), with_host_lengths(
        torch.tensor([700, 700], dtype=torch.int64, device=dev), [700, 700]))
    assert int(cpb[1]) == int(cpb1[0]) == int(cpb1[1]) and int(cpb[1]) % 8 == 0
    assert torch.equal(child1[0], child1[1])
    assert grown.shape[1] == P + child.shape[1] and n_after.tolist() == [P + int(cpb[0]), 700 + int(cpb[1])]
