"""Farthest-point sampling: grid-wide (cooperative) form vs one workgroup per cloud, same samples.
usage: python tools/fps_bench.py"""
import os, subprocess, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tools_common import timeit
from iso_points_amd.point_processing import farthest_sampling
dev = torch.device("cuda:0")
one = bool(os.environ.get("ISO_FPS_ONE_WORKGROUP"))
for P, ns in ((20000, 2000), (50000, 5000), (500000, 5000), (1000000, 10000)):
    g = torch.Generator().manual_seed(P)
    pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
    num = torch.tensor([P], device=dev)
    ratio = ns / P
    t = timeit(lambda: farthest_sampling(pts, num, ratio), warm=1, rep=3)
    idx = farthest_sampling(pts, num, ratio)[2]
    print("%s P=%d samples=%d: %.2f ms (%.2f us/sample) checksum %d" % ("one-workgroup" if one else "default", P, ns, t,
          t * 1e3 / ns, int(idx.sum())), flush=True)
