"""Entry points added in round 3, each against the path it replaces or against torch:
iso_splat_tile_offsets (vs iso_prefix_sum), iso_bricks_build_whole (vs iso_points_bbox + iso_bricks_build),
iso_splat_view_mask_scan + iso_splat_front_rows (vs iso_splat_view_mask + iso_splat_front), the median radius in pieces
with the histograms of a pass summed over two "ranks" (vs torch.median of the whole), sticky grid counters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(P, seed, dev):
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    return (p + 0.02 * (torch.rand(P, 3, generator=g) - 0.5)).to(dev).contiguous()


@pytest.mark.parametrize("n", [1, 7, 4097, 70000])
def test_tile_offsets_is_the_prefix_sum_and_clears_its_input(dev, n):
    from iso_points_amd import _lib
    g = torch.Generator().manual_seed(n)
    cnt = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32).to(dev)
    ref = torch.cumsum(cnt.long(), 0) - cnt.long()
    off = torch.full((n,), -7, dtype=torch.int32, device=dev)
    cur = torch.full((n,), -7, dtype=torch.int32, device=dev)
    p = _lib.ptr
    _lib.call("iso_splat_tile_offsets", p(cnt), p(off), p(cur), n, _lib.stream())
    assert torch.equal(off.long(), ref) and (cur == 0).all() and (cnt == 0).all()


@pytest.mark.parametrize("P,knn", [(30000, 8), (5000, 4), (257, 8)])
def test_build_whole_equals_bbox_plus_build(dev, P, knn):
    """The grid that takes its own bounding box = the grid built from iso_points_bbox, and a second build on the same
    workspace (counters left zeroed by the offsets pass, accumulators reset by their reader) gives the same again."""
    from iso_points_amd import bricks
    pts = _cloud(P, 3, dev)
    nrm = torch.nn.functional.normalize(pts, dim=-1).contiguous()
    a, b = bricks.BrickGrid(P, dev), bricks.BrickGrid(P, dev)
    a.build(pts, nrm, knn_k=knn)                                               # iso_bricks_build_whole
    b.build(pts, nrm, bbox=bricks.points_bbox(pts), knn_k=knn)                 # explicit box -> iso_bricks_build
    ha, hb = a.header(), b.header()
    for k in ("f", "r", "inv_sigma", "diag", "nb", "n_bricks", "n", "occupied"):
        assert ha[k] == hb[k], k
    ma, ia, da = bricks.resample_fused(a, knn + 1, want_idx=True)
    mb, ib, db = bricks.resample_fused(b, knn + 1, want_idx=True)
    assert torch.equal(ma, mb) and torch.equal(ia, ib) and torch.equal(da, db)
    a.build(pts, nrm, knn_k=knn)
    m2, i2, _ = bricks.resample_fused(a, knn + 1, want_idx=True)
    assert torch.equal(m2, ma) and torch.equal(i2, ia)
    # the counters of the earlier grids live on in the sticky block: one read reports all of them
    c = a.counters_since_last()
    assert c[0] == 2 * ha["occupied"] and a.counters_since_last() == [0] * 16


def test_view_mask_scan_and_front_rows_equal_the_five_launch_path(dev):
    from iso_points_amd import bricks
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    P, N = 20011, 3
    pts = _cloud(P, 5, dev)
    nrm = torch.nn.functional.normalize(pts, dim=-1).contiguous()
    views = torch.stack([look_at_view(3.0, 15.0, 100.0 * i) for i in range(N)]).to(dev).contiguous()
    projs = (views @ perspective(30.0).to(dev)).contiguous()
    m0, c0 = bricks.view_mask(pts, nrm, views)
    m1, c1, scanned = bricks.view_mask_scan(pts, nrm, views)
    assert torch.equal(m0, m1) and torch.equal(c0[:N], c1[:N])
    per_view = ((m0[None] >> torch.arange(N, device=dev)[:, None]) & 1).sum(1)
    assert torch.equal(scanned[2], per_view.long()) and torch.equal(scanned[1], torch.cumsum(per_view, 0) - per_view)
    ss = SurfaceSplatting(raster_settings=PointsRasterizationSettings(image_size=64, points_per_pixel=4))
    grid = bricks.BrickGrid(P, dev)
    grid.build(pts, nrm, payload=m0, radius=0.2, cell_scale=bricks.H_CELL_SCALE)
    h = bricks.splat_h_fused(grid, m0, c0, N)
    a = ss.front_setup(pts, nrm, views, projs, m0, h, features_from_normals=True)                       # iso_splat_front
    b = ss.front_setup(pts, nrm, views, projs, m1, h, features_from_normals=True, scanned=scanned)      # iso_splat_front_rows
    tot = int(per_view.sum())
    assert torch.equal(a["first_idx"], b["first_idx"]) and torch.equal(a["num_points"], b["num_points"])
    for k in ("ndc", "ellipse_params", "radii", "scaler", "features", "src", "cutoff_threshold"):
        assert torch.equal(a[k][:tot], b[k][:tot]), k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_median_in_pieces_over_two_ranks_equals_torch_median(dev, seed):
    """Two 'ranks' hold disjoint row ranges; each counts its own visible rows per pass, the pass's histograms are summed
    (what the all-reduce does) before the next pass: the result is the lower median of ALL visible radii, as
    torch.median gives it (rasterizer.py:884), for every cloud."""
    from iso_points_amd import _lib
    g = torch.Generator().manual_seed(seed)
    N, lens = 3, [5000, 1, 777]
    P = sum(lens)
    radii = (torch.rand(P, 2, generator=g) * 0.05).to(dev).contiguous()
    radii[10] = radii[11]                                                        # ties
    vis = (torch.rand(P, generator=g) < 0.6).to(torch.uint8).to(dev)
    if seed == 2:
        vis[lens[0]:lens[0] + 1] = 0                                             # a cloud without a visible point -> 0
    first = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int64, device=dev)
    num = torch.tensor(lens, dtype=torch.int64, device=dev)
    lib = _lib.load()
    nb = lib.iso_splat_median_radius_workspace_bytes(N)
    words = lib.iso_splat_median_pass_words(N)
    p = _lib.ptr
    # rank A: the first half of every cloud's rows, rank B: the rest (as sub-ranges of the same packed arrays)
    half = [l // 2 for l in lens]
    fa, na = first.clone(), torch.tensor(half, dtype=torch.int64, device=dev)
    fb, nbr = first + na, num - na
    wa = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    wb = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    for ps in range(3):
        for (f_, n_, w_) in ((fa, na, wa), (fb, nbr, wb)):
            _lib.call("iso_splat_median_pass", ps, p(radii), p(vis), p(f_), p(n_), N, max(lens), p(w_), nb, _lib.stream())
        sa = wa[:12 * words].view(torch.int32)[ps * words:(ps + 1) * words]
        sb = wb[:12 * words].view(torch.int32)[ps * words:(ps + 1) * words]
        tot = sa + sb
        sa.copy_(tot); sb.copy_(tot)
    out = torch.empty((N,), dtype=torch.float32, device=dev)
    _lib.call("iso_splat_median_final", p(wa), N, 10.0, p(out), _lib.stream())
    assert (wa == 0).all()                                                       # left zeroed
    for n in range(N):
        rows = radii[int(first[n]):int(first[n]) + lens[n]][vis[int(first[n]):int(first[n]) + lens[n]].bool()]
        ref = rows.reshape(-1).median().item() * 10.0 if rows.numel() else 0.0
        assert out[n].item() == pytest.approx(ref, rel=0, abs=0) or abs(out[n].item() - ref) <= 1e-7 * max(ref, 1e-30)


@pytest.mark.parametrize("S,density", [(40, "sparse"), (40, "dense"), (50, "dense"), (64, "ring"), (37, "sparse")])
def test_occupancy_backward_over_gradient_densities(dev, S, density):
    """The heavy-point kernel walks pixel masks of the gradient image (eight lanes per point, a row path for blocks that are
    fully covered, a bit-scan path otherwise): sparse / dense / ring-shaped gradient images, sides that are and are not
    multiples of 8 and of 4, two clouds, both supports (disc = rasterize_points_backward.cu:156, rectangle =
    rasterize_points.cu:726-746) against the oracle."""
    from oracle import splat_oracle as SO
    from test_splat_gpu import sphere_scene
    from iso_points_amd.rasterizer import _C
    from util import rel_err
    sc = sphere_scene(2500, n_views=2, S=S, seed=91)
    g = torch.Generator().manual_seed(S)
    go = torch.randn(2, S, S, generator=g)
    if density == "sparse":
        go[torch.rand(2, S, S, generator=g) > 0.02] = 0.0
    elif density == "ring":
        ax = (torch.arange(S) + 0.5) / S * 2 - 1
        rr = (ax[None, :] ** 2 + ax[:, None] ** 2).sqrt()
        go[:, (rr - 0.6).abs() > 0.04] = 0.0
    rs = torch.tensor([0.35, 0.2])
    ref = SO.occ_backward(sc["ndc"], sc["radii"], go, sc["first"], sc["num"], 10.0, rs=rs, mode=2)
    got = _C._splat_points_occ_fast_cuda_backward(sc["ndc"].to(dev), sc["radii"].to(dev), rs.to(dev), go.to(dev),
                                                  sc["num"].to(dev), sc["first"].to(dev))
    assert ref.abs().sum() > 0 and rel_err(got, ref) < 1e-6
    again = _C._splat_points_occ_fast_cuda_backward(sc["ndc"].to(dev), sc["radii"].to(dev), rs.to(dev), go.to(dev),
                                                    sc["num"].to(dev), sc["first"].to(dev))
    assert torch.equal(got, again)                                    # bit-stable from run to run
    ref_r = SO.occ_backward(sc["ndc"], sc["radii"], go, sc["first"], sc["num"], 6.0, mode=1)
    got_r = _C._splat_points_occ_backward(sc["ndc"].to(dev), sc["radii"].to(dev), go.to(dev), sc["first"].to(dev),
                                          sc["num"].to(dev), 6.0, 0.05)
    assert rel_err(got_r, ref_r) < 1e-6
