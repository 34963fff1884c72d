import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from iso_points_amd.sdf_models import Siren
from util import sphere_cloud
dev = torch.device("cuda:0")
for H, L, P in ((128, 2, 150001), (256, 3, 100000)):
    torch.manual_seed(0)
    m = Siren(hidden_size=H, n_layers=L).to(dev)
    pts = sphere_cloud(P, seed=3).to(dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8)
    ref = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    bad = 0
    for rep in range(20):
        out = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
        d = (out.points != ref.points).any(-1).sum().item()
        bad += d > 0
    print("H=%d: %d of 20 repeats differ (last: %d points)" % (H, bad, d))
