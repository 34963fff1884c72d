"""Determinism / position-independence checks of the fused SIREN kernels (eval and Newton)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
import bench
from iso_points_amd.sdf_models import siren_sdf_and_grad
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
dev = torch.device("cuda:0")
from iso_points_amd.sdf_models import Siren
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sys.argv) > 1:
    torch.manual_seed(0)
    model = Siren(hidden_size=H, n_layers=L).to(dev)       # random weights (chaotic Newton: any bit flips show)
else:
    model = bench.fitted_siren(dev, steps=100)
g = torch.Generator().manual_seed(1)
for P in (200000, 20011):
    pts = bench.sphere_cloud(P, seed=3, device=dev)
    s1, g1 = siren_sdf_and_grad(model, pts[0])
    perm = torch.randperm(P, generator=g).to(dev)
    s2, g2 = siren_sdf_and_grad(model, pts[0][perm].contiguous())
    print("P=%d eval permutation-invariant:" % P, torch.equal(s1[perm], s2), torch.equal(g1[perm], g2))
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
    r1 = proj._project_points(model, pts, full_lengths(pts), proj_max_iters=10)
    for rep in range(3):
        r2 = proj._project_points(model, pts, full_lengths(pts), proj_max_iters=10)
        print("  project repeat %d: points %s normals %s" % (rep, torch.equal(r1.points, r2.points), torch.equal(r1.normals, r2.normals)))
    half = pts[:, : P // 2].contiguous()
    r3 = proj._project_points(model, half, full_lengths(half), proj_max_iters=10)
    d = (r1.points[:, : P // 2] - r3.points).abs().max().item()
    print("  first half alone == first half of whole:", torch.equal(r1.points[:, : P // 2], r3.points), d)
