"""How many points of the cfg-3a cycle reach k_splat_backward_heavy (per cycle), the search radius in pixels and the
share of gradient-carrying 8x8 blocks.  usage: python tools/diag/heavy_count.py
(measured with counters in the first form of the kernel: 3.7 flagged-block visits and 18.4 gradient pixels per heavy point)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd import _lib
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev) if len(sys.argv) > 1 and sys.argv[1] == "siren" else SphereSDF().to(dev)
cyc = bench.Cycle(dev, model, Comm(enabled=False))
cyc.cyc.use_graphs = False
hip = ctypes.CDLL("libamdhip64.so")
seen = []
real_call = _lib.call


def spy(name, *a):
    rc = real_call(name, *a)
    if name == "iso_splat_backward":
        # (pts, radii, visible, rs, first, num, N, max_pts, go, idx, gz, S, W, K, rect, radii_s, P, ws, ws_bytes, grad, stream)
        torch.cuda.synchronize()
        N, S, W, P, ws, ws_b = a[6], a[11], a[12], a[16], a[17].value, a[18]
        maps = ws_b - (64 + (4 * P + 15) // 16 * 16 + 16 + 8 * P)
        cnt, rs = ctypes.c_int32(0), ctypes.c_float(0)
        hip.hipMemcpy(ctypes.byref(cnt), ctypes.c_void_p(ws + maps), 4, 2)
        hip.hipMemcpy(ctypes.byref(rs), a[3], 4, 2)
        nby, nbx = (S + 7) // 8, (W + 7) // 8
        nb2x = (nbx + 7) // 8
        rows = (ctypes.c_uint8 * (N * nby * nb2x))()                      # row bytes: one bit per 8x8 block (k_grad_maps)
        hip.hipMemcpy(rows, ctypes.c_void_p(ws + 8 * N * nby * nbx), N * nby * nb2x, 2)
        flagged = sum(bin(b).count("1") for b in rows)
        pm = (ctypes.c_uint64 * (N * nby * nbx))()
        hip.hipMemcpy(pm, ctypes.c_void_p(ws), 8 * N * nby * nbx, 2)
        print("gradient pixels per view:", [sum(bin(x).count("1") for x in pm[v * nby * nbx:(v + 1) * nby * nbx]) for v in range(N)])
        seen.append((P, cnt.value, rs.value, S, flagged / (N * nby * nbx)))
    return rc


_lib.call = spy
cyc.step()
torch.cuda.synchronize()
P, cnt, rs, S, fl = seen[-1]
print("rows %d, heavy points %d (%.1f %%), search radius %.5f ndc = %.1f px of %d, flagged 8x8 blocks %.1f %%"
      % (P, cnt, 100.0 * cnt / max(P, 1), rs, rs * S / 2, S, 100.0 * fl))
