"""Where a workgroup of k_raster spends its time (needs a library built with -DRS_DBG_PHASES:
tools/build_variant.sh rphases splat.hip -DRS_DBG_PHASES): shader-clock cycles of thread 0 between the marks, summed over all
workgroups of one cfg-3a cycle.  usage: ISO_DEV_LIB=tools/variants/libiso_rphases.so python tools/diag/raster_phases.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd import _lib
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
dev = torch.device("cuda:0")
siren = "siren" in sys.argv[1:]        # the headline cycle instead of cfg 3a
cyc = bench.Cycle(dev, bench.fitted_siren(dev) if siren else SphereSDF().to(dev), Comm(enabled=False))
cyc.cyc.use_graphs = False
lib = _lib.load()
buf = (ctypes.c_double * 16)()
names = ["stage the chunk's records + barrier", "candidates -> hit lists (+ barrier)", "pixels insert their hits (thread 0)",
         "barrier after the insertions (other pixels)", "q of the survivors", "outputs / slice scratch (+ compositing)"]
cyc.step(); torch.cuda.synchronize()
lib.iso_dbg_raster_phases(buf)
cyc.step(); torch.cuda.synchronize()
lib.iso_dbg_raster_phases(buf)
tot = sum(buf[:8])
print("k_raster -- %d work items, %.0f cycles each" % (buf[8], tot / max(buf[8], 1)))
for i in range(6):
    print("  %-50s %6.1f %%  %8.2f Mcycles" % (names[i], 100 * buf[i] / max(tot, 1), buf[i] / 1e6))
