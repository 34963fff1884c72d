"""How many queries of the cfg-3a cycle go to the tail kernels (per cycle), and the distribution of the K-th neighbour
distance in units of the search radius r.  usage: python tools/diag/tail_counts.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
from iso_points_amd import bricks
dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, SphereSDF().to(dev), Comm(enabled=False))
cyc.cyc.use_graphs = False
cyc.cyc.grid.counters_since_last()
out = cyc.step()
u = cyc.cyc.usage(out[4])
print("per cycle:", {k: u["grid"][k] for k in ("occupied", "tail", "tail_h", "overflow_bricks")})
# K-th distance distribution of the resample stage's input
c = cyc.cyc
r0 = c.proj._project_points(c.model, c.pts0_local, c.num_local, proj_max_iters=10)
pts, nrm = r0.points[0].contiguous(), r0.normals[0].contiguous()
g = bricks.BrickGrid(pts.shape[0], dev)
g.build(pts, nrm, knn_k=8, cell_scale=0.8 * 8)
moved, idx, d2 = bricks.resample_fused(g, 9, want_idx=True)
hd = g.header()
r = hd["r"]
dk = d2[:, -1].clamp_min(0).sqrt() / r
valid = d2[:, -1] >= 0
print("r = %.5f, f = %.5f; neighbours found for %.4f of the points" % (r, hd["f"], valid.float().mean().item()))
for q in (0.5, 0.9, 0.99, 0.999):
    print("d_K / r quantile %.3f: %.3f" % (q, torch.quantile(dk[valid][:2000000].float(), q).item()))
for t in (0.5, 0.55, 0.6, 0.65, 0.7, 0.8):
    print("fraction with d_K > %.2f r: %.5f" % (t, ((dk > t) | ~valid).float().mean().item()))
