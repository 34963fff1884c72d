"""Only the fused kernels, for PMC passes: build + resample (cell 0.8 r) + h (cell 6 spacings), 1 M points."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd.bricks import BrickGrid, points_bbox, resample_fused, splat_h_fused, view_mask
from iso_points_amd.cameras import look_at_view

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev).contiguous()
nrm = pts.clone()
views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(4)]).to(dev).contiguous()
grid = BrickGrid(P, dev)
bbox = points_bbox(pts)
mask, cnt = view_mask(pts, nrm, views)
for _ in range(6):
    grid.build(pts, nrm, bbox=bbox, knn_k=8)
    resample_fused(grid, 9)
    grid.build(pts, nrm, payload=mask, bbox=bbox, radius=0.2, cell_scale=6.0)
    splat_h_fused(grid, mask, cnt, 4)
torch.cuda.synchronize()
print(grid.header())
