"""Stage timeline of the point-stationary SIREN step kernel (variant library built with -DPS_DBG_TIMES):
   tools/build_variant.sh psdbg siren_ps.hip "-DPS_DBG_TIMES"; ISO_SIREN_PS=1 ISO_DEV_LIB=tools/variants/libiso_psdbg.so python tools/ps_stage_times.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.sdf_models import PackedSiren, Siren  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Siren(hidden_size=256, n_layers=3).to(dev)
ps = PackedSiren(m, dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
pts = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1).to(dev).contiguous()
sdf = torch.empty((P,), device=dev)
grad = torch.empty((P, 3), device=dev)
ws = ps.workspace(P)
for _ in range(3):
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), ps.hidden,
              ps.n_hidden, ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
lib = _lib.load()
assert lib.iso_dbg_ps_times(buf) == 0
if os.environ.get("PS_INNER"):
    for w in range(4):
        t = [buf[w * 64 + i] for i in range(45)]
        print("wave %d:" % w, " ".join(str(t[i + 1] - t[i]) for i in range(44)))
    sys.exit(0)
names = ["points+layer0", "fwd0", "fwd1", "fwd2(top)", "rev2", "rev1", "rev0", "epilogue"]
for w in range(4):
    t = [buf[w * 64 + i] for i in range(9)]
    d = [t[i + 1] - t[i] for i in range(8)]
    print("wave %d: total %6d | " % (w, t[8] - t[0]) + "  ".join("%s %d" % (n, x) for n, x in zip(names, d)))
