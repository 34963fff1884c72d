"""Length of the active list every Newton launch of the headline cycle consumes (the fitted SIREN, 1 M points):
counts[it] of the projection workspace after project(T = 10) and after the resample stage's project(T = 3).
usage: python tools/diag/newton_counts.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd import _lib
from iso_points_amd.dist import Comm
dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, bench.fitted_siren(dev), Comm(enabled=False))
cyc.cyc.use_graphs = False
hip = ctypes.CDLL("libamdhip64.so")
lib = _lib.load()
real = _lib.call


def spy(name, *a):
    rc = real(name, *a)
    if name == "iso_project_siren":
        # (pts, out, normals, mask, n, packed, H, L, w0, wh, T, tol, ws, ws_bytes, stream)
        torch.cuda.synchronize()
        n, H, L, T, ws = a[4], a[6], a[7], a[10], a[12].value
        stash = lib.iso_project_siren_workspace_bytes(n, H, L) - (2 * n * 4 + 64 * 4 + 64)
        buf = (ctypes.c_int32 * 16)()
        hip.hipMemcpy(buf, ctypes.c_void_p(ws + stash + 8 * n), 64, 2)
        print("project T=%d of %d points: active list lengths %s" % (T, n, [n] + list(buf[1:T + 2])))
    return rc


cyc.step(); torch.cuda.synchronize()
_lib.call = spy
cyc.step(); torch.cuda.synchronize()
