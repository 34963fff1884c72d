"""One cycle alone, for kernel traces (no CPU baseline): the cfg-3a cycle (analytic sphere SDF), or with `siren` the
headline cycle (cfg 3b: the SIREN of bench.fitted_siren).
usage: python tools/cycle_only.py [steps] [siren]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
import bench
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
siren = "siren" in sys.argv[2:]
cyc = bench.Cycle(dev, bench.fitted_siren(dev) if siren else SphereSDF().to(dev), Comm(enabled=False))
cyc.cyc.marks = False          # as bench.analytic_cycle: no SDF kernel to bracket
for _ in range(2):
    cyc.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    cyc.step()
torch.cuda.synchronize()
print(("cfg3b" if siren else "cfg3a") + " cycle: %.3f ms" % ((time.perf_counter() - t0) / steps * 1e3))
