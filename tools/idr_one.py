import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from oracle import iso_oracle as O
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.sdf_models import idr_sdf_and_grad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6).to(dev)
pts = (torch.nn.functional.normalize(torch.randn(300000, 3), dim=-1) * 0.6).to(dev)
for _ in range(4):
    idr_sdf_and_grad(m, pts)
torch.cuda.synchronize()
from tools_common import timeit
print("IDR 8x512 300k: %.2f ms" % timeit(lambda: idr_sdf_and_grad(m, pts), warm=1, rep=5))
