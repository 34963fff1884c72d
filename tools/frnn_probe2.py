import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from iso_points_amd import frnn
from iso_points_amd.dist import Comm
from iso_points_amd.levelset_sampling import with_host_lengths
from tools_common import timeit
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
C = bench.Cycle(dev, model, Comm(enabled=False)); cyc = C.cyc
r1 = cyc.project_resample()
pts, nrm = r1.points[0], r1.normals[0]
ss = cyc.splat
flags, off, lens = ss.filter_renderable(pts, nrm, cyc.views)
tot = sum(lens); N = 4
pts_f = ss.compact(pts, flags, off, pts.shape[0], tot)
mx = max(lens)
fl = [sum(lens[:i]) for i in range(N)]
padded = torch.zeros((N, mx, 3), device=dev)
for i in range(N):
    padded[i, :lens[i]] = pts_f[fl[i]:fl[i] + lens[i]]
num = with_host_lengths(torch.tensor(lens, device=dev), lens)
r4 = torch.full((4,), 0.2, device=dev)
grid = frnn.build_grid(padded, num, r4)
print("grid params", grid.params.tolist())
print("bench filtered clouds K=7: %.3f ms" % timeit(lambda: frnn.frnn_grid_points(padded, padded, num, num, K=7, r=r4, grid=grid)))
d, i, _, g2 = frnn.frnn_grid_points(padded, padded, num, num, K=7, r=r4, grid=grid)
print("tail counts", g2.tail_counts.tolist())
for n in range(N):
    dd = d[n, :lens[n]]
    print(n, "found<7:", (dd[:, 6] < 0).sum().item(), "max 7th d2", dd[:, 6].max().item(), "mean", dd[:, 6].mean().item())
# occupancy of cells
cnt = (grid.off[0, 1:int(grid.params[0,7].item())] - grid.off[0, :int(grid.params[0,7].item())-1])
print("cells nonempty", (cnt > 0).sum().item(), "max per cell", cnt.max().item(), "mean nonempty", cnt[cnt>0].float().mean().item())
