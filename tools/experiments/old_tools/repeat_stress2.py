import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths, SphereTracing
from iso_points_amd.sdf_models import Siren, siren_sdf_and_grad
from oracle import iso_oracle as O
from util import sphere_cloud
dev = torch.device("cuda:0")
# value-only kernels + tracing, SIREN H=256 / H=128, IDR 5x256
for H, L in ((256, 3), (128, 2)):
    torch.manual_seed(H)
    m = Siren(hidden_size=H, n_layers=L).to(dev)
    pts = sphere_cloud(200000, seed=5)[0].to(dev)
    ref = siren_sdf_and_grad(m, pts, need_grad=False)[0]
    refg = siren_sdf_and_grad(m, pts)
    bad = 0
    for rep in range(200):
        v = siren_sdf_and_grad(m, pts, need_grad=False)[0]
        s, g = siren_sdf_and_grad(m, pts)
        bad += int(not (torch.equal(v, ref) and torch.equal(s, refg[0]) and torch.equal(g, refg[1])))
    print("SIREN H=%d value-only + full evaluation, 200k points: %d of 200 repeats differ" % (H, bad))
torch.manual_seed(2)
idr = O.IdrSDF(hidden_size=256, n_layers=5, skip_in=(), num_frequencies=4)
with torch.no_grad():
    for prm in idr.parameters():
        prm.add_(0.02 * torch.randn_like(prm))
idr = idr.to(dev)
x = ((torch.rand(1, 150000, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 2).to(dev)
pr = UniformProjection(proj_max_iters=5, proj_tolerance=1e-30, knn_k=8)
ref = pr._project_points(idr, x, full_lengths(x), proj_max_iters=5)
bad = 0
for rep in range(100):
    out = pr._project_points(idr, x, full_lengths(x), proj_max_iters=5)
    bad += int(not (torch.equal(out.points, ref.points) and torch.equal(out.normals, ref.normals)))
print("IDR 5x256, 150k points, T=5: %d of 100 repeats differ" % bad)
