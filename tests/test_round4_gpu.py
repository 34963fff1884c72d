"""Round 4: fused launches, each against the separate passes it replaces (bit for bit).
  iso_project_sphere_follow + iso_bricks_build_pending = iso_project_sphere + iso_bricks_build_whole
                               + iso_splat_view_mask_scan (bounding box left pending in the grid workspace, header made
                               by the count pass, renderable mask taken by the projection, its scan riding in the count launch)
  the one-launch brick offsets scan (inside every build) = the two-launch scan
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(P, seed, dev, jitter=0.05):
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    return (p + jitter * (torch.rand(P, 3, generator=g) - 0.5)).to(dev).contiguous()


def _views(dev, n):
    from iso_points_amd.cameras import look_at_view
    return torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(n)]).to(dev).contiguous()


@pytest.mark.parametrize("P,T,n_views", [(1, 10, 1), (1023, 3, 2), (1025, 10, 4), (50000, 3, 4), (300007, 10, 3)])
def test_project_sphere_follow_equals_the_separate_passes(dev, P, T, n_views):
    from iso_points_amd import bricks
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    pts = _cloud(P, P, dev).view(1, P, 3)
    m = SphereSDF().to(dev)
    proj = UniformProjection()
    ref = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=T)
    views = _views(dev, n_views)
    # the separate passes on the projected cloud
    rp, rn = ref.points[0].contiguous(), ref.normals[0].contiguous()
    mask_ref, tot_ref, scanned_ref = bricks.view_mask_scan(rp, rn, views, 1.0, 100.0, True)
    g_ref = bricks.BrickGrid(P, dev)
    g_ref.build(rp, rn, payload=mask_ref, radius=0.2, cell_scale=bricks.H_CELL_SCALE)
    # the same with the side work done by the projection launch, twice on one workspace (the pending box must be back
    # at its rest state)
    grid = bricks.BrickGrid(P, dev)
    for rep in range(2):
        f = bricks.Follow(grid, P, views=views)
        res = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=T, follow=f)
        assert f.done
        assert torch.equal(res.points, ref.points) and torch.equal(res.normals, ref.normals) and torch.equal(res.mask, ref.mask)
        assert torch.equal(f.mask, mask_ref)
        if rep == 0:                                       # the pending box as 8 floats; taking it clears it
            assert torch.equal(bricks.box_take(grid), bricks.points_bbox(rp))
            proj._project_points(m, pts, full_lengths(pts), proj_max_iters=T, follow=f)
        grid.build(rp, rn, payload=f.mask, radius=0.2, cell_scale=bricks.H_CELL_SCALE, pending=True, follow=f)
        assert torch.equal(f.total, tot_ref)
        n_chunks = (P + 1023) // 1024                      # (the workspace is sized for 8 views: the rest is not written)
        assert torch.equal(f.scanned[0].view(torch.int32)[:n_views * n_chunks], scanned_ref[0].view(torch.int32)[:n_views * n_chunks])
        for a, b in zip(f.scanned[1:], scanned_ref[1:]):
            assert torch.equal(a, b)
        ha, hb = grid.header(), g_ref.header()
        for k in ("f", "r", "inv_sigma", "diag", "nb", "n_bricks", "n", "occupied"):
            assert ha[k] == hb[k], (k, ha[k], hb[k])
        h1 = bricks.splat_h_fused(grid, f.mask, f.total, n_views)
        h2 = bricks.splat_h_fused(g_ref, mask_ref, tot_ref, n_views)
        sel = (mask_ref[None, :] >> torch.arange(n_views, device=dev)[:, None]) & 1
        assert torch.equal(h1[sel.bool()], h2[sel.bool()])


def test_follow_header_only_feeds_the_resample_grid(dev):
    """header + box for the resample grid (no mask part): fused kernel results equal the stand-alone build's."""
    from iso_points_amd import bricks
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import SphereSDF
    P, knn = 40001, 8
    pts = _cloud(P, 9, dev).view(1, P, 3)
    m = SphereSDF().to(dev)
    proj = UniformProjection()
    grid = bricks.BrickGrid(P, dev)
    f = bricks.Follow(grid, P)
    r = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10, follow=f)
    rp, rn = r.points[0].contiguous(), r.normals[0].contiguous()
    grid.build(rp, rn, knn_k=knn, pending=True)
    ref = bricks.BrickGrid(P, dev).build(rp, rn, knn_k=knn)
    a = bricks.resample_fused(grid, knn + 1, want_idx=True)
    b = bricks.resample_fused(ref, knn + 1, want_idx=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_cycle_with_follow_equals_cycle_without(dev):
    """IsoCycle (one rank, analytic SDF): the fused side work changes nothing in any output of the cycle."""
    import bench
    from iso_points_amd.dist import Comm
    from iso_points_amd.sdf_models import SphereSDF
    from iso_points_amd import bricks
    old = bench.P_TOTAL
    bench.P_TOTAL = 60000
    try:
        outs = []
        for use in (True, False):
            cyc = bench.Cycle(dev, SphereSDF().to(dev), Comm(enabled=False))
            cyc.cyc.marks = False
            cyc.cyc.use_graphs = False
            if not use:
                cyc.cyc.no_follow = True
            outs.append(cyc.step())
            torch.cuda.synchronize()
    finally:
        bench.P_TOTAL = old
    (r1a, imga, ga, fa, fra), (r1b, imgb, gb, fb, frb) = outs
    assert torch.equal(r1a.points, r1b.points) and torch.equal(r1a.normals, r1b.normals)
    assert torch.equal(imga, imgb) and torch.equal(fa.idx, fb.idx) and torch.equal(fa.zbuf, fb.zbuf)
    assert torch.equal(fra["first_idx"], frb["first_idx"]) and torch.equal(fra["num_points"], frb["num_points"])
    rows = int((fra["first_idx"][-1] + fra["num_points"][-1]).item())      # (rows beyond the clouds are not written)
    assert rows > 0 and torch.equal(ga[:rows], gb[:rows])


@pytest.mark.parametrize("fitted,P,T", [(False, 150001, 10), (True, 200003, 10), (True, 70001, 3), (False, 999, 4)])
def test_newton_tail_in_one_launch_equals_a_launch_per_iteration(dev, fitted, P, T):
    """k_siren_tail_x3 (iterations k..T of the projection in one launch, workgroups iterating the survivors of their own
    tiles) against the launch-per-iteration form: points, normals, masks and the per-iteration active counts, bit for
    bit, for chaotic random weights (lists stay long: the tail takes over long lists) and for a fitted network (lists
    die out), from every iteration the tail can start at."""
    from iso_points_amd import _lib
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    from iso_points_amd.sdf_models import Siren
    from oracle import iso_oracle as O
    lib = _lib.load()
    torch.manual_seed(0)
    m = Siren(hidden_size=256, n_layers=3).to(dev)
    if fitted:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        g = torch.Generator(device="cpu").manual_seed(0)
        for _ in range(150):
            x = ((torch.rand(4096, 3, generator=g) - 0.5) * 3.0).to(dev)
            loss = ((m(x).sdf - (x.norm(dim=-1, keepdim=True) - 1.0)) ** 2).mean()
            opt.zero_grad(); loss.backward(); opt.step()
    for prm in m.parameters():
        prm.requires_grad_(False)
    pts = _cloud(P, 3, dev).view(1, P, 3)
    proj = UniformProjection()
    n_off = lib.iso_project_siren_workspace_bytes(P, 256, 3) - 64 * 4 - 64

    def run(k):
        lib.iso_siren_set_tail_from(k)
        try:
            r = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=T)
            torch.cuda.synchronize()
            counts = proj._packed_cache._ws[n_off:n_off + 64 * 4].view(torch.int32)[1:T + 1].tolist()
        finally:
            lib.iso_siren_set_tail_from(-1)
        return r, counts

    ref, c_ref = run(0)
    for k in [-1] + list(range(1, T + 1)):
        out, c = run(k)
        assert torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals) and torch.equal(out.mask, ref.mask), k
        assert c == c_ref, (k, c, c_ref)
