"""Repeatability stress test of the fused SIREN Newton projection (bit-exact across repeats and
against a shuffled copy).  usage: python tools/siren_stress.py H L P repeats"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
import bench
from iso_points_amd.sdf_models import Siren
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
H, L, P, R = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Siren(hidden_size=H, n_layers=L).to(dev)
pts = bench.sphere_cloud(P, seed=3, device=dev)
proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
ref = proj._project_points(model, pts, full_lengths(pts), proj_max_iters=10)
bad = 0
g = torch.Generator().manual_seed(7)
for r in range(R):
    if r % 2 == 0:
        out = proj._project_points(model, pts, full_lengths(pts), proj_max_iters=10)
        ok = torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals)
    else:
        perm = torch.randperm(P, generator=g).to(dev)
        out = proj._project_points(model, pts[:, perm].contiguous(), full_lengths(pts), proj_max_iters=10)
        ok = torch.equal(out.points, ref.points[:, perm]) and torch.equal(out.normals, ref.normals[:, perm])
    bad += 0 if ok else 1
print("H=%d L=%d P=%d: %d of %d repeats differ" % (H, L, P, bad, R))
