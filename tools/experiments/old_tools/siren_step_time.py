"""Median / minimum time of one Newton projection (T moves, T + 1 launches with the move epilogue) of the bench's 1 M-point
cloud on the fitted SIREN.  usage: python tools/siren_step_time.py [T=1] [repeats=40]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
pts = bench.sphere_cloud(bench.P_TOTAL, seed=0, device=dev)
num = full_lengths(pts)
proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
for _ in range(5):
    proj._project_points(model, pts, num, proj_max_iters=T)
torch.cuda.synchronize()
ts = []
for _ in range(rep):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); proj._project_points(model, pts, num, proj_max_iters=T); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("%-24s T=%d  median %.3f ms  min %.3f ms  (%d repeats)" % (os.path.basename(os.environ.get("ISO_DEV_LIB", "default")), T, ts[len(ts) // 2], ts[0], rep))
