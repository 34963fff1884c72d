"""Long repeatability stress of the shipped MFMA kernels (operand loads are pinned between MFMAs in the
8-wave shapes): N repeats of a chaotic-weight projection, every bit compared.  usage: python tools/repeat_stress.py [N]"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from iso_points_amd.sdf_models import Siren
from oracle import iso_oracle as O
from util import sphere_cloud
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Siren(hidden_size=256, n_layers=3).to(dev)
pts = sphere_cloud(300000, seed=3).to(dev)
proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8)
ref = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
bad = 0
for rep in range(N):
    out = proj._project_points(m, pts, full_lengths(pts), proj_max_iters=10)
    bad += int(not (torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals)))
print("SIREN 4x256, 300k points, T=10: %d of %d repeats differ" % (bad, N))
torch.manual_seed(1)
idr = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6)
with torch.no_grad():
    for prm in idr.parameters():
        prm.add_(0.02 * torch.randn_like(prm))
idr = idr.to(dev)
x = ((torch.rand(1, 100000, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 2).to(dev)
pr = UniformProjection(proj_max_iters=5, proj_tolerance=1e-30, knn_k=8)
ref = pr._project_points(idr, x, full_lengths(x), proj_max_iters=5)
bad = 0
for rep in range(N // 4):
    out = pr._project_points(idr, x, full_lengths(x), proj_max_iters=5)
    bad += int(not (torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals)))
print("IDR 8x512, 100k points, T=5: %d of %d repeats differ" % (bad, N // 4))
