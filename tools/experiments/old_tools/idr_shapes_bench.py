"""IDR eval throughput for several shapes (ISO_DEV_LIB selects an alternative build)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from tools_common import timeit
from oracle import iso_oracle as O   # model definition only
from iso_points_amd.sdf_models import idr_sdf_and_grad
dev = torch.device("cuda:0")
for H, NL, skip, NF, P in [(512, 8, (4,), 6, 300000), (256, 5, (), 4, 500000), (128, 3, (1,), 0, 1000000), (256, 8, (4,), 6, 500000)]:
    torch.manual_seed(0)
    m = O.IdrSDF(hidden_size=H, n_layers=NL, skip_in=skip, num_frequencies=NF).to(dev)
    pts = (torch.nn.functional.normalize(torch.randn(P, 3), dim=-1) * 0.6).to(dev)
    t = timeit(lambda: idr_sdf_and_grad(m, pts), warm=1, rep=5)
    print("IDR %dx%d skip %s F=%d: %d pts %.2f ms  %.2f Mevals/s" % (NL, H, skip, NF, P, t, P / t / 1e3))
