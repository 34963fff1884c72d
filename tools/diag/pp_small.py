import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from tools_common import timeit
from iso_points_amd import point_processing as pp
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for P in (5000, 24000, 100000):
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
    num = torch.tensor([P], device=dev)
    t = timeit(lambda: pp.wlop(p, num, ratio=0.5, neighborhood_size=16, iters=3, perturb=False), warm=1, rep=3)
    print("wlop P %6d ratio 0.5 iters 3: %7.2f ms" % (P, t), flush=True)
    t = timeit(lambda: pp.upsample(p, P + P // 2, num_points=num, neighborhood_size=16), warm=1, rep=3)
    print("upsample P %6d -> %d: %7.2f ms" % (P, P + P // 2, t), flush=True)
