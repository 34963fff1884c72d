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

# IDR-style network (split-fp16 kernel for H = 256 / 512): perturbed weights, fixed iteration count
sys.path.insert(0, "/root/repo")
from oracle import iso_oracle as O   # model definition only
for H, NL, skip, NF, P in ((512, 8, (4,), 6, 60000), (256, 5, (), 4, 100000)):
    torch.manual_seed(H)
    m = O.IdrSDF(hidden_size=H, n_layers=NL, skip_in=skip, num_frequencies=NF)
    with torch.no_grad():
        for prm in m.parameters():
            prm.add_(0.02 * torch.randn_like(prm))
    m = m.to(dev)
    pts = ((torch.rand(1, P, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 2).to(dev)
    proj = UniformProjection(proj_max_iters=6, proj_tolerance=1e-30, knn_k=8)
    ref = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=6)
    bad = 0
    for rep in range(10):
        out = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=6)
        d = ((out.points != ref.points).any(-1) | (out.normals != ref.normals).any(-1)).sum().item()
        bad += d > 0
    print("IDR %dx%d: %d of 10 repeats differ (last: %d points)" % (NL, H, bad, d))
