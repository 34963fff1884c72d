import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from tools_common import timeit
from iso_points_amd.point_processing import farthest_sampling
dev = torch.device("cuda:0")
P = 500000
g = torch.Generator().manual_seed(P)
pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
num = torch.tensor([P], device=dev)
prev = None
for ns in (64, 128, 256, 512, 1024, 2048, 5000, 10000, 20000):
    t = timeit(lambda: farthest_sampling(pts, num, ns / P), warm=1, rep=3)
    print("samples %6d: %8.2f ms%s" % (ns, t, "" if prev is None else "  marginal %.2f us/sample" % ((t - prev[1]) * 1e3 / (ns - prev[0]))), flush=True)
    prev = (ns, t)
