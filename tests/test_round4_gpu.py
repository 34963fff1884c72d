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
    n_off = lib.iso_project_siren_counts_offset(P, 256, 3)

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
    assert 0 < c_ref[0] <= P and all(x >= y for x, y in zip(c_ref, c_ref[1:])), c_ref      # (the counters, not some other words)
    for k in [-1] + list(range(1, T + 1)):
        out, c = run(k)
        assert torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals) and torch.equal(out.mask, ref.mask), k
        assert c == c_ref, (k, c, c_ref)


def test_front_rows_respects_the_row_capacity(dev):
    """rows beyond `capacity` are dropped and reported, never written (the arrays end there); the rows that
    fit are the rows of the unbounded call, and the 'visible' flags of the rows come back cleared."""
    from iso_points_amd import bricks
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    P, N = 30011, 3
    pts = _cloud(P, 5, dev, jitter=0.02)
    nrm = torch.nn.functional.normalize(pts, dim=-1).contiguous()
    views = _views(dev, N)
    projs = (views @ perspective(30.0).to(dev)).contiguous()
    ss = SurfaceSplatting(raster_settings=PointsRasterizationSettings(image_size=128, points_per_pixel=8))
    mask, cnt, scanned = bricks.view_mask_scan(pts, nrm, views)
    grid = bricks.BrickGrid(P, dev).build(pts, nrm, payload=mask, radius=0.2, cell_scale=bricks.H_CELL_SCALE)
    h = bricks.splat_h_fused(grid, mask, cnt, N)
    full = ss.front_setup(pts, nrm, views, projs, mask, h, features_from_normals=True, scanned=scanned)
    rows = int(full["num_points"].sum().item())
    assert int(full["row_overflow"].item()) == 0 and (full["visible"][:rows] == 0).all()
    cap = rows - 1234
    guard = torch.full((12 * cap + 4096,), 7.0, device=dev)
    small = ss.front_setup(pts, nrm, views, projs, mask, h, features_from_normals=True, scanned=scanned, out=guard, capacity=cap)
    torch.cuda.synchronize()
    assert int(small["row_overflow"].item()) == 1
    assert (guard[12 * cap:] == 7.0).all()
    # whole views that fit are identical; the view that is cut is identical up to the cut
    first = full["first_idx"].tolist()
    for k in ("ndc", "ellipse_params", "radii", "scaler"):
        v0 = min(int(first[1]), cap)
        assert torch.equal(small[k][:v0], full[k][:v0]), k
    ss._row_overflow = None


@pytest.mark.parametrize("S,bin_size,P,K", [(64, 16, 3000, 8), (100, 32, 20000, 5), (256, 16, 40000, 8), (33, 8, 500, 3),
                                            (64, 16, 6000, 64), (48, 16, 4000, 150)])   # K > 32: lists in the output arrays
def test_rasterize_fine_of_coarse_is_splat_points(dev, S, bin_size, P, K):
    """DSS._C._rasterize_coarse / _rasterize_fine (ext.cpp:11-12): the bin table against a numpy restatement of
    rasterize_points.cu:341-385 (same float32 expressions), and fine(coarse(x)) == splat_points(x) bit for bit."""
    import numpy as np
    from iso_points_amd.rasterizer import _C
    from splat_util import random_splats
    sc = random_splats(P, N=2, seed=S + P)
    pts, el, cut, rad, first, num = (sc[k].to(dev) for k in ("ndc", "ellipse", "cutoff", "radii", "first", "num"))
    M = int(num.max().item())
    bins = _C._rasterize_coarse(pts, rad, first, num, S, bin_size, M)
    B = 1 + (S - 1) // bin_size
    assert bins.shape == (2, B, B, M) and bins.dtype == torch.int32
    # numpy restatement (float32 throughout)
    p, r = pts.cpu().numpy().astype(np.float32), rad.cpu().numpy().astype(np.float32)
    f32 = np.float32
    half = f32(1.0) / f32(S)
    def ndc(i):
        return f32(-1) + (f32(2 * i) + f32(1.0)) / f32(S)
    got = bins.cpu().numpy()
    for n in range(2):
        a, b = int(first[n]), int(first[n] + num[n])
        px0, px1 = p[a:b, 0] - r[a:b, 0], p[a:b, 0] + r[a:b, 0]
        py0, py1 = p[a:b, 1] - r[a:b, 1], p[a:b, 1] + r[a:b, 1]
        front = ~(p[a:b, 2] < 0)
        for by in range(B):
            y0, y1 = ndc(by * bin_size) - half, ndc((by + 1) * bin_size - 1) + half
            for bx in range(B):
                x0, x1 = ndc(bx * bin_size) - half, ndc((bx + 1) * bin_size - 1) + half
                hit = front & (py0 <= y1) & (y0 <= py1) & (px0 <= x1) & (x0 <= px1)
                want = (np.nonzero(hit)[0] + a).astype(np.int32)
                row = got[n, by, bx]
                assert (row[:len(want)] == want).all() and (row[len(want):] == -1).all(), (n, by, bx)
    ref = _C.splat_points(pts, el, cut, rad, first, num, 0.05, S, K, 0, 0)
    out = _C._rasterize_fine(pts, el, cut, rad, bins, 0.05, S, bin_size, K)
    for a, b, name in zip(out, ref, ("idx", "zbuf", "qvalue", "occupancy")):
        assert torch.equal(a, b), name
    # the reference's limits: too many bins per side, too many points in a bin
    with pytest.raises(RuntimeError):
        _C._rasterize_coarse(pts, rad, first, num, 512, 16, 8)
    with pytest.raises(RuntimeError):
        _C._rasterize_coarse(pts, rad, first, num, S, bin_size, 3)


def _h_fused_vs_standalone(dev, pts, nrm, n_views, cell_scale=None, image_size=64):
    """splat_h_fused against the K = 7 query of each filtered view cloud + vrk_h (rasterizer.py:256-300 of the
    reference); returns the grid header."""
    from iso_points_amd.bricks import BrickGrid, H_CELL_SCALE, splat_h_fused, view_mask
    from iso_points_amd.cameras import look_at_view
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from iso_points_amd.levelset_sampling import with_host_lengths
    P = pts.shape[0]
    views = torch.stack([look_at_view(3.0, 20.0, 360.0 / n_views * i) for i in range(n_views)]).to(dev).contiguous()
    ss = SurfaceSplatting(raster_settings=PointsRasterizationSettings(image_size=image_size))
    flags, off, lens = ss.filter_renderable(pts, nrm, views)
    tot = sum(lens)
    mask, cnt = view_mask(pts, nrm, views)
    grid = BrickGrid(P, dev).build(pts, nrm, payload=mask, radius=ss.frnn_radius,
                                   cell_scale=H_CELL_SCALE if cell_scale is None else cell_scale)
    h = splat_h_fused(grid, mask, cnt, n_views)
    first = [sum(lens[:i]) for i in range(n_views)]
    num = with_host_lengths(torch.tensor(lens, dtype=torch.int64, device=dev), lens)
    fst = with_host_lengths(torch.tensor(first, dtype=torch.int64, device=dev), first)
    ss.per_point_info(ss.compact(pts, flags, off, P, tot), ss.compact(nrm, flags, off, P, tot), fst, num, views, views)
    fl = flags[:-1].view(n_views, P).bool()
    got = torch.cat([h[v][fl[v]] for v in range(n_views)])
    bad = (got != ss._Vrk_h).nonzero().flatten()
    assert bad.numel() == 0, (bad.numel(), got[bad[:8]].tolist(), ss._Vrk_h[bad[:8]].tolist())
    return grid.header()


def test_h_tail_stray_points_at_every_distance(dev):
    """The bandwidth tail search keeps exact distances only below 0.02 (where the clamp of vrk_h still lets them through)
    and otherwise asks whether ANY renderable point lies in [0.02, r^2): stray points at distances on both sides of
    sqrt(0.02) = 0.1414 and of r = 0.2 from a dense sphere, stray pairs / triples whose mutual distances straddle the
    same marks, far from anything else."""
    g = torch.Generator().manual_seed(41)
    base = torch.nn.functional.normalize(torch.randn(60000, 3, generator=g), dim=-1)
    stray = []
    for k, d in enumerate([0.02, 0.05, 0.1, 0.13, 0.1405, 0.1414, 0.1416, 0.142, 0.15, 0.19, 0.1999, 0.2, 0.2001, 0.25, 0.4]):
        u = torch.nn.functional.normalize(torch.randn(40, 3, generator=g), dim=-1)
        stray.append(u * (1.0 + d))
        stray.append(u[:10] * (1.0 - d))
    far = torch.tensor([[3.0, 0.0, 0.0], [0.0, 3.0, 0.5], [0.0, -3.0, 0.5], [2.0, 2.0, 2.0]])
    groups = []
    for c, d in zip(far, [0.05, 0.141, 0.142, 0.199]):                  # pairs and a triple, isolated from the sphere
        groups += [c[None], c[None] + torch.tensor([[d, 0.0, 0.0]]), c[None] + torch.tensor([[0.0, 0.21, 0.0]])]
    pts = torch.cat([base] + stray + groups)
    pts = pts[torch.randperm(pts.shape[0], generator=g)]
    nrm = torch.nn.functional.normalize(pts, dim=-1)
    hdr = _h_fused_vs_standalone(dev, pts.to(dev).contiguous(), nrm.to(dev).contiguous(), 4)
    assert hdr["tail_h"] > 100, hdr


def test_h_tail_of_overfull_bricks(dev):
    """Bricks too full to stage send all their queries to the tail kernel, whose four waves then share thousands of
    records per ring: every record must be visited exactly once whatever each wave has found so far."""
    g = torch.Generator().manual_seed(42)
    base = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    clump = torch.tensor([[0.0, 0.0, 1.0]]) + 0.004 * torch.randn(6000, 3, generator=g)       # 6000 points in one brick
    clump2 = torch.tensor([[0.6, 0.0, 0.8]]) + 0.02 * torch.randn(9000, 3, generator=g)
    pts = torch.cat([base, clump, clump2])
    pts = pts[torch.randperm(pts.shape[0], generator=g)]
    nrm = torch.nn.functional.normalize(pts, dim=-1)
    hdr = _h_fused_vs_standalone(dev, pts.to(dev).contiguous(), nrm.to(dev).contiguous(), 3)
    assert hdr["overflow_bricks"] > 0 and hdr["tail_h"] > 1000, hdr


def test_brick_workspace_check_tells_an_uninitialised_workspace(dev):
    """iso_bricks_workspace_check: ISO_ERR_INVALID until iso_bricks_workspace_init has run on the workspace."""
    from iso_points_amd import _lib
    lib = _lib.load()
    n = 5000
    nbytes = int(lib.iso_bricks_workspace_bytes(n))
    ws = torch.full((nbytes,), 0x55, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError):
        _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n, _lib.stream())
    _lib.call("iso_bricks_workspace_init", _lib.ptr(ws), n, _lib.stream())
    _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n, _lib.stream())


def test_brick_workspace_check_tells_a_dirty_workspace_and_small_clouds_find_the_scan_words(dev):
    """(a) the check also refuses a workspace whose arrival words / scan totals are not zero (what an aborted
    build leaves behind); (b) the zero-on-entry chunk totals of the one-launch brick scan sit at a fixed offset, so a
    build of n < n_max points on a workspace initialised for n_max (iso_bricks_build_whole carves with n) finds them:
    its grid equals the grid of a workspace initialised for exactly n."""
    from iso_points_amd import _lib, bricks
    from util import sphere_cloud
    lib = _lib.load()
    n_max, n = 60000, 7000
    ws = torch.full((int(lib.iso_bricks_workspace_bytes(n_max)),), 0x55, dtype=torch.uint8, device=dev)
    _lib.call("iso_bricks_workspace_init", _lib.ptr(ws), n_max, _lib.stream())
    _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n_max, _lib.stream())
    pts = sphere_cloud(n, seed=2)[0].to(dev).contiguous()
    nrm = torch.nn.functional.normalize(pts, dim=-1).contiguous()
    small = torch.full((int(lib.iso_bricks_workspace_bytes(n)),), 0x55, dtype=torch.uint8, device=dev)
    _lib.call("iso_bricks_workspace_init", _lib.ptr(small), n, _lib.stream())
    for w in (ws, small):
        _lib.call("iso_bricks_build_whole", _lib.ptr(pts), _lib.ptr(nrm), None, n, -1.0, 8, bricks.RESAMPLE_CELL * 8, _lib.ptr(w),
                  w.numel(), _lib.stream())
    outs = []
    for w in (ws, small):
        out = torch.empty_like(pts)
        _lib.call("iso_resample_fused", _lib.ptr(w), n, _lib.ptr(pts), n, 9, _lib.ptr(out), None, None, _lib.stream())
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n_max, _lib.stream())
    # a dirty arrival word (counter block: int 40 at byte 256) / a dirty scan total (right behind the 576-int block)
    for byte in (256 + 4 * 40, 256 + 4 * 576):
        saved = ws[byte:byte + 4].clone()
        ws[byte:byte + 4] = 1
        with pytest.raises(RuntimeError):
            _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n_max, _lib.stream())
        ws[byte:byte + 4] = saved
    _lib.call("iso_bricks_workspace_check", _lib.ptr(ws), n_max, _lib.stream())
