"""FPS sequences of the library on six cloud types, saved to a file (one process per mode: the mode switches are read once):
   ISO_FPS_LAZY=0 python tools/diag/fps_ab.py /tmp/a.pt ; python tools/diag/fps_ab.py /tmp/b.pt ; python tools/diag/fps_ab.py cmp /tmp/a.pt /tmp/b.pt
plus the time of 5 000 of 500 k."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    ok = True
    for k in a:
        same = torch.equal(a[k], b[k])
        ok = ok and same
        print("%-28s %s (%d samples)" % (k, "identical" if same else "DIFFERENT at %d" % int((a[k] != b[k]).nonzero()[0, -1]), a[k].shape[-1]))
    sys.exit(0 if ok else 1)
from tools_common import timeit
from iso_points_amd.point_processing import farthest_sampling
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
sph = lambda n: torch.nn.functional.normalize(torch.randn(1, n, 3, generator=g), dim=-1)
lat = torch.stack(torch.meshgrid(*([torch.arange(42.0)] * 3), indexing="ij"), -1).view(1, -1, 3) * 0.05
clumps = torch.cat([torch.randn(1, 30000, 3, generator=g) * 0.05, torch.randn(1, 30000, 3, generator=g) * 0.05 + 4.0], 1)
dup = sph(40000); dup[0, 20000:] = dup[0, :20000]
lat14 = torch.stack(torch.meshgrid(*([torch.arange(14.0)] * 3), indexing="ij"), -1).view(1, -1, 3) * 0.1
dup3 = sph(3000); dup3[0, 1500:] = dup3[0, :1500]
cases = {"sphere 70 k / 2000": (sph(70000), 2000), "sphere 500 k / 5000": (sph(500000), 5000), "lattice 42^3 / 3000 (ties)": (lat, 3000),
         "volume 200 k / 4000": (torch.rand(1, 200000, 3, generator=g), 4000), "two clumps 60 k / 1500": (clumps, 1500),
         "duplicates 40 k / 25000": (dup, 25000), "sphere 1.2 M / 3000": (sph(1200000), 3000), "all of 9 k": (sph(9000), 9000),
         # below 8 k points: the one-workgroup kernels (register-resident k_fps_reg; ISO_FPS_ONE_WORKGROUP=1: k_fps)
         "sphere 5 k / 2500": (sph(5000), 2500), "lattice 14^3 / all (ties)": (lat14, 2744), "sphere 777 / 500": (sph(777), 500),
         "duplicates 3 k / 2500": (dup3, 2500), "sphere 8191 / 4000": (sph(8191), 4000)}
out = {}
for k, (p, ns) in cases.items():
    P = p.shape[1]
    out[k] = farthest_sampling(p.to(dev), torch.tensor([P], device=dev), ns / P)[2].cpu()
    assert out[k].shape[1] >= ns - 1 and (k.startswith("dup") or len(set(out[k][0].tolist())) == out[k].shape[1]), k
    print(k, "done", flush=True)
torch.save(out, sys.argv[1])
p = cases["sphere 500 k / 5000"][0].to(dev); num = torch.tensor([500000], device=dev)
print("5000 of 500 k: %.2f ms" % timeit(lambda: farthest_sampling(p, num, 0.01), warm=1, rep=3))
