"""Distribution of candidates per 16x16 tile in the bench's splat forward (load balance of k_raster)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF

dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, SphereSDF().to(dev), Comm(enabled=False))
r = cyc.cyc.project_resample()
keep = []
_zeros = torch.zeros
def zeros(*a, **k):
    t = _zeros(*a, **k)
    if t.dtype == torch.int32 and t.dim() == 1 and t.numel() == 4 * 32 * 32 + 1:
        keep.append(t)
    return t
torch.zeros = zeros
frags, filt = cyc.cyc.splat_forward(r.points[0], r.normals[0])
torch.cuda.synchronize()
torch.zeros = _zeros
c = keep[0][:-1].float().cpu()          # first such tensor = tile_cnt (the second is the cursor)
print("tiles %d  pairs %d  mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  nonzero %d" % (
    c.numel(), int(c.sum()), c.mean(), c.median(), c.quantile(0.9), c.quantile(0.99), c.max(), int((c > 0).sum())))
srt = c.sort(descending=True).values
print("top 16:", [int(x) for x in srt[:16]])
print("share of pairs in top 1%% tiles: %.2f, top 10%%: %.2f" % (srt[: c.numel() // 100].sum() / c.sum(), srt[: c.numel() // 10].sum() / c.sum()))
