"""Launch-by-launch cost of iso_project_siren on the bench's fitted SIREN: T = 0, 1, 2, 3, 10."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths

dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
pts = bench.sphere_cloud(bench.P_TOTAL, seed=0, device=dev)
num = full_lengths(pts)
proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)

def timeit(T, rep=5):
    for _ in range(2):
        proj._project_points(model, pts, num, proj_max_iters=T)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); proj._project_points(model, pts, num, proj_max_iters=T); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]

prev = 0.0
for T in (0, 1, 2, 3, 4, 10):
    t = timeit(T)
    print("T=%2d  %.3f ms  (+%.3f)" % (T, t, t - prev))
    prev = t
