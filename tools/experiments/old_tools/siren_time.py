"""project(T=1) and project(T=10) of the bench's 1 M points with whatever kernel the environment selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
pts = bench.sphere_cloud(bench.P_TOTAL, seed=0, device=dev)
num = full_lengths(pts)
proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
out = []
for T in (1, 10):
    for _ in range(3):
        proj._project_points(model, pts, num, proj_max_iters=T)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); proj._project_points(model, pts, num, proj_max_iters=T); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    out.append("T=%d %.3f ms" % (T, ts[3]))
print(os.environ.get("TAG", ""), "  ".join(out), flush=True)
