"""Where a workgroup of the fused brick kernels spends its time (needs a library built with -DBK_DBG_PHASES:
tools/build_variant.sh phases bricks.hip -DBK_DBG_PHASES): shader-clock cycles of thread 0 between the marks, summed over
all workgroups of one cfg-3a cycle, for the bandwidth kernel and the resample kernel separately.
usage: ISO_DEV_LIB=tools/variants/libiso_phases.so python tools/diag/brick_phases.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd import _lib
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, SphereSDF().to(dev), Comm(enabled=False))
cyc.cyc.use_graphs = False
lib = _lib.load()
buf = (ctypes.c_double * 16)()
names = ["run bounds + prefix", "record loads + cell counts", "cell scan", "payload loads + scatter + query runs",
         "queries of wave 0 (h: pass 1)", "h: pass 2 / resample: wait for the other waves", "entry barrier (previous brick / start)",
         "resample: wide-window queries"]
real = _lib.call


def spy(name, *a):
    if name in ("iso_resample_fused", "iso_splat_h_fused"):
        torch.cuda.synchronize(); lib.iso_dbg_brick_phases(buf)
    rc = real(name, *a)
    if name in ("iso_resample_fused", "iso_splat_h_fused"):
        torch.cuda.synchronize(); lib.iso_dbg_brick_phases(buf)
        tot = sum(buf[:8])
        print(name, "-- %d workgroups with work, %.0f cycles each" % (buf[8], tot / max(buf[8], 1)))
        for i in range(8):
            print("  %-70s %6.1f %%  %8.2f Mcycles" % (names[i], 100 * buf[i] / max(tot, 1), buf[i] / 1e6))
    return rc


cyc.step()
torch.cuda.synchronize()
_lib.call = spy
cyc.step()
torch.cuda.synchronize()
