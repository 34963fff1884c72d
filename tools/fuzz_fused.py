"""Fuzz of the fused neighbour kernels against the stand-alone paths (which are pinned to the oracle and the reference's
golden vectors by tests/): many adversarial clouds -- clusters of every density, stray points, exact duplicates, planes,
lines, lattices with massive distance ties, mixtures -- through
  * resample_fused  vs  frnn.frnn_grid_points + UniformProjection.repulsion_step   (neighbour lists, d2, moves: bit-exact)
  * splat_h_fused   vs  the K = 7 query of every filtered view cloud + vrk_h        (bit-exact)
usage: python tools/fuzz_fused.py [n_cases] [first_seed]      exit code 1 on any mismatch (the seed is printed)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")


def make_cloud(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    parts = []
    kind = seed % 7
    P0 = int(200 + r(1).item() ** 2 * 60000)
    if kind in (0, 1, 2, 5, 6):
        parts.append(torch.nn.functional.normalize(n(P0, 3), dim=-1) * (0.5 + r(1)) + 0.02 * r(1) * n(P0, 3))
    if kind in (1, 4, 5):                                   # clusters of every density
        for _ in range(int(1 + r(1).item() * 6)):
            m = int(10 + r(1).item() ** 2 * 8000)
            parts.append(n(1, 3) * 0.7 + n(m, 3) * (10.0 ** (-1 - 3 * r(1).item())))
    if kind in (2, 4, 6):                                   # stray points and stray groups
        m = int(5 + r(1).item() * 400)
        parts.append(n(m, 3) * 1.5)
        for _ in range(4):
            c = n(1, 3) * 2.0
            parts.append(c + n(int(2 + r(1).item() * 9), 3) * (0.02 + 0.2 * r(1)))
    if kind in (3, 5):                                      # plane, line, lattice (massive ties)
        m = int(500 + r(1).item() * 20000)
        q = r(m, 3)
        q[:, 2] = 0.3
        parts.append(q)
        t = r(int(50 + r(1).item() * 3000), 1)
        parts.append(torch.cat([t, 0.5 * t, -t], dim=1))
        k = int(4 + r(1).item() * 14)
        parts.append(torch.stack(torch.meshgrid(*([torch.arange(float(k))] * 3), indexing="ij"), -1).view(-1, 3) * (0.01 + 0.1 * r(1)))
    if kind in (3, 6):                                      # exact duplicates
        base = torch.cat(parts) if parts else r(100, 3)
        pick = (r(int(10 + r(1).item() * 500)) * base.shape[0]).long().clamp(max=base.shape[0] - 1)
        parts.append(base[pick])
    pts = torch.cat(parts)
    pts = pts[torch.randperm(pts.shape[0], generator=g)]
    if pts.shape[0] > 150000:
        pts = pts[:150000]
    nrm = torch.nn.functional.normalize(pts + 0.3 * n(pts.shape[0], 3), dim=-1)
    return pts.contiguous(), nrm.contiguous()


def check_resample(pts, nrm, knn_k):
    from iso_points_amd import frnn
    from iso_points_amd.bricks import BrickGrid, resample_fused
    from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
    P = pts.shape[0]
    grid = BrickGrid(P, dev).build(pts, nrm, knn_k=knn_k)
    out, idx, d2 = resample_fused(grid, knn_k + 1, want_idx=True)
    hdr = grid.header()
    num = full_lengths(pts[None])
    dists, idxs, _, _ = frnn.frnn_grid_points(pts[None], pts[None], num, num, K=knn_k + 1, r=hdr["r"])
    bad = []
    if not torch.equal(idx, idxs[0, :, 1:]):
        bad.append("neighbour lists (%d rows)" % (idx != idxs[0, :, 1:]).any(dim=1).sum().item())
    if not torch.equal(d2, dists[0, :, 1:]):
        bad.append("d2")
    inv_sigma = torch.tensor([hdr["inv_sigma"]], dtype=torch.float32, device=dev)
    ref = UniformProjection(knn_k=knn_k).repulsion_step(pts[None], nrm[None], idxs[..., 1:], inv_sigma)[0]
    same = torch.equal(out.view(torch.int32), ref.view(torch.int32)) or (
        torch.equal(out.isnan(), ref.isnan()) and torch.equal(out.nan_to_num(), ref.nan_to_num()))
    if not same:
        bad.append("moves (%d rows)" % (out.nan_to_num() != ref.nan_to_num()).any(dim=1).sum().item())
    return bad, hdr


def check_h(pts, nrm, n_views, seed):
    from iso_points_amd.bricks import BrickGrid, H_CELL_SCALE, splat_h_fused, view_mask
    from iso_points_amd.cameras import look_at_view
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from iso_points_amd.levelset_sampling import with_host_lengths
    P = pts.shape[0]
    views = torch.stack([look_at_view(3.0 + 0.1 * (seed % 5), 20.0, 360.0 / n_views * i + seed) for i in range(n_views)]).to(dev).contiguous()
    ss = SurfaceSplatting(raster_settings=PointsRasterizationSettings(image_size=64, backface_culling=bool(seed % 3)))
    flags, off, lens = ss.filter_renderable(pts, nrm, views)
    tot = sum(lens)
    mask, cnt = view_mask(pts, nrm, views, backface_culling=bool(seed % 3))
    if cnt[:n_views].tolist() != lens:
        return ["view counts %s vs %s" % (cnt[:n_views].tolist(), lens)], {}
    grid = BrickGrid(P, dev).build(pts, nrm, payload=mask, radius=ss.frnn_radius,
                                   cell_scale=H_CELL_SCALE if seed % 4 else 0.5)
    h = splat_h_fused(grid, mask, cnt, n_views)
    hdr = grid.header()
    if tot == 0:
        return [], hdr
    first = [sum(lens[:i]) for i in range(n_views)]
    num = with_host_lengths(torch.tensor(lens, dtype=torch.int64, device=dev), lens)
    fst = with_host_lengths(torch.tensor(first, dtype=torch.int64, device=dev), first)
    ss.per_point_info(ss.compact(pts, flags, off, P, tot), ss.compact(nrm, flags, off, P, tot), fst, num, views, views)
    fl = flags[:-1].view(n_views, P).bool()
    got = torch.cat([h[v][fl[v]] for v in range(n_views)])
    nbad = (got != ss._Vrk_h).sum().item()
    return (["h (%d of %d rows)" % (nbad, got.numel())] if nbad else []), hdr


def run(n_cases, seed0):
    """-> number of mismatching cases (each printed with its seed)"""
    failures = 0
    tails = [0, 0, 0]
    for seed in range(seed0, seed0 + n_cases):
        pts, nrm = make_cloud(seed)
        pts, nrm = pts.to(dev), nrm.to(dev)
        knn_k = (4, 8, 12)[seed % 3]
        n_views = (1, 3, 4, 8)[seed % 4]
        for name, fn in (("resample K=%d" % knn_k, lambda: check_resample(pts, nrm, knn_k)),
                         ("h views=%d" % n_views, lambda: check_h(pts, nrm, n_views, seed))):
            try:
                bad, hdr = fn()
            except Exception as e:                           # a capacity the path refuses is a finding too
                bad, hdr = ["raised %r" % (e,)], {}
            tails[0] += hdr.get("tail", 0)
            tails[1] += hdr.get("tail_h", 0)
            tails[2] += hdr.get("overflow_bricks", 0)
            if bad:
                failures += 1
                print("MISMATCH seed %d P %d %s: %s  %s" % (seed, pts.shape[0], name, "; ".join(bad), hdr), flush=True)
    print("fuzz_fused: %d cases, %d mismatches; tail queries %d, tail (query, view) pairs %d, overfull bricks %d"
          % (n_cases, failures, tails[0], tails[1], tails[2]))
    return failures


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1000) else 0)
