"""Stage times of the fused neighbour kernels against the stand-alone path (1 M-point sphere)."""
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import frnn
from iso_points_amd.bricks import BrickGrid, H_CELL_SCALE, points_bbox, resample_fused, splat_h_fused, view_mask
from iso_points_amd.cameras import look_at_view
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev).contiguous()
    nrm = pts.clone()
    views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(4)]).to(dev).contiguous()
    out = {"P": P}
    grid = BrickGrid(P, dev)
    out["bbox_ms"] = timeit(lambda: points_bbox(pts))
    bbox = points_bbox(pts)
    for cs in (0.7, 0.8, 0.9, 1.0):
        out["build_ms_cell%.1f" % cs] = timeit(lambda: grid.build(pts, nrm, bbox=bbox, knn_k=8, cell_scale=cs * 8))
        out["resample_fused_ms_cell%.1f" % cs] = timeit(lambda: resample_fused(grid, 9))
        out["hdr_cell%.1f" % cs] = grid.header()
    grid.build(pts, nrm, bbox=bbox, knn_k=8)
    out["resample_fused_idx_ms"] = timeit(lambda: resample_fused(grid, 9, want_idx=True))
    num = full_lengths(pts[None])
    proj = UniformProjection(knn_k=8)
    out["standalone_tree_ms"] = timeit(lambda: proj._create_tree(pts[None], True, num), n=10)
    inv_sigma = torch.tensor([grid.header()["inv_sigma"]], device=dev)
    out["standalone_repulse_ms"] = timeit(lambda: proj.repulsion_step(pts[None], nrm[None], proj._knn_idx, inv_sigma))
    out["view_mask_ms"] = timeit(lambda: view_mask(pts, nrm, views))
    mask, cnt = view_mask(pts, nrm, views)
    for cs in (5.0, 6.0, 7.0):
        out["h_build_ms_cell%.1f" % cs] = timeit(lambda: grid.build(pts, nrm, payload=mask, bbox=bbox, radius=0.2, cell_scale=cs))
        out["h_fused_ms_cell%.1f" % cs] = timeit(lambda: splat_h_fused(grid, mask, cnt, 4))
        out["h_hdr_cell%.1f" % cs] = grid.header()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
