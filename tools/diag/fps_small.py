"""FPS at the reference's working sizes (wlop: half of 5 k .. 50 k points): ms and us per sample.  One process per mode
(ISO_FPS_LAZY / ISO_FPS_PPT / ISO_FPS_ONE_WORKGROUP are read once or per call): python tools/diag/fps_small.py"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from tools_common import timeit
from iso_points_amd.point_processing import farthest_sampling
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for P in (2500, 5000, 8192, 12000, 24000, 50000, 100000, 200000):
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
    num = torch.tensor([P], device=dev)
    ns = P // 2 if P <= 50000 else 5000
    t = timeit(lambda: farthest_sampling(p, num, ns / P), warm=1, rep=3)
    print("P %7d samples %6d: %8.2f ms  %.2f us/sample" % (P, ns, t, t * 1e3 / ns), flush=True)
