import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import frnn
from iso_points_amd.levelset_sampling import full_lengths, with_host_lengths
from tools_common import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
P = 1000000
pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
num = full_lengths(pts)
r7 = torch.full((1,), 0.2, device=dev)
def q(p, n, K, r, **kw):
    grid = frnn.build_grid(p, n, r)
    t = timeit(lambda: frnn.frnn_grid_points(p, p, n, n, K=K, r=r, grid=grid, **kw))
    return t
print("full sphere 1M  K=7 r=.2 : %.3f ms" % q(pts, num, 7, r7))
hemi = pts[:, pts[0, :, 2] > 0].contiguous()
nh = full_lengths(hemi)
print("hemisphere %d K=7 r=.2 : %.3f ms" % (hemi.shape[1], q(hemi, nh, 7, r7)))
# 4 rotated hemispheres, padded batch
hs = []
for ax, sgn in ((0, 1), (0, -1), (2, 1), (2, -1)):
    hs.append(pts[0, sgn * pts[0, :, ax] > 0])
mx = max(h.shape[0] for h in hs)
pad = torch.zeros(4, mx, 3, device=dev)
lens = [h.shape[0] for h in hs]
for i, h in enumerate(hs):
    pad[i, :lens[i]] = h
n4 = with_host_lengths(torch.tensor(lens, device=dev), lens)
r4 = torch.full((4,), 0.2, device=dev)
print("4 hemispheres batch K=7  : %.3f ms" % q(pad, n4, 7, r4))
tr = (torch.sqrt(torch.tensor([2 * 3 ** 0.5 / P])) * 8).to(dev)
print("full sphere 1M  K=9 tree : %.3f ms" % q(pts, num, 9, tr))
print("full sphere 1M  K=9 tree +nn : %.3f ms" % q(pts, num, 9, tr, return_nn=True))
