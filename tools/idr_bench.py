import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from tools_common import timeit
from iso_points_amd import _lib as _L
if os.environ.get("ISO_DEV_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from oracle import iso_oracle as O   # model definition only (weights); compute is the HIP path
from iso_points_amd.sdf_models import idr_sdf_and_grad, PackedIdr
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6).to(dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
g = torch.Generator().manual_seed(0)
pts = (torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1) * 0.6).to(dev)
t = timeit(lambda: idr_sdf_and_grad(m, pts[0]), warm=1, rep=5)
mac = 39*512 + 6*512*512 + 473*512 + 512      # forward MACs (a3: ~1.84 M)
flop = 2 * 2 * mac * P
print("IDR 8x512 eval %d pts: %.2f ms  -> %.1f TFLOP/s (fwd+bwd MACs), %.2f Mpts/s" % (P, t, flop / t / 1e9, P / t / 1e3))
proj = UniformProjection()
num = full_lengths(pts)
t = timeit(lambda: proj._project_points(m, pts, num, proj_max_iters=10), warm=1, rep=3)
r = proj._project_points(m, pts, num, proj_max_iters=10)
print("IDR project T=10: %.2f ms, converged %.3f" % (t, r.mask.float().mean().item()))
