"""When the workgroups of k_siren_step_x3_both end, per XCD (a library built with -DX3_DBG_END: tools/build_variant.sh x3end
siren_x3.hip "-DX3_DBG_END").  Static tile assignment gives every CU the same number of tiles; the XCDs do not run at the
same clock under the power cap, so the launch ends with the slowest one.
usage: ISO_DEV_LIB=tools/variants/libiso_x3end.so python tools/diag/x3_end_times.py [P]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from iso_points_amd import _lib  # noqa: E402
from iso_points_amd.sdf_models import PackedSiren, Siren  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Siren(hidden_size=256, n_layers=3).to(dev)
g = torch.Generator().manual_seed(0)
pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev).contiguous()
ps = PackedSiren(m, dev)
sdf = torch.empty((P,), dtype=torch.float32, device=dev)
grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
ws = ps.workspace(P)
lib = ctypes.CDLL(os.path.abspath(os.environ["ISO_DEV_LIB"]))


def run():
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), ps.hidden,
              ps.n_hidden, ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())


for warm in (3, 400, 400):                   # cold clocks, then the power-limited steady state twice
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    nb = 512
    buf = (ctypes.c_ulonglong * (3 * nb))()
    assert lib.iso_dbg_x3_end(buf, nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 3).astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    dur = (a[:, 1] - t0) / 100.0             # us (100 MHz clock)
    start = (a[:, 0] - t0) / 100.0
    print("after %d warm-up passes: %d workgroups, launch %.1f us (first start to last end)" % (warm, len(a), dur.max()))
    for x in sorted(set(a[:, 2].tolist())):
        sel = a[:, 2] == x
        big = sel & (np.arange(len(a)) < 256)
        print("  XCD %d: %3d workgroups  start %.1f..%.1f  end of the 96-point workgroups: mean %.1f  min %.1f  max %.1f us;"
              "  all: max %.1f" % (x, sel.sum(), start[sel].min(), start[sel].max(), dur[big].mean() if big.any() else 0,
                                   dur[big].min() if big.any() else 0, dur[big].max() if big.any() else 0, dur[sel].max()))
