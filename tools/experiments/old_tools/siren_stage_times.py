"""Stage timeline of the split-bf16 SIREN step kernel (needs a library built with -DX3_DBG_TIMES:
tools/build_variant.sh times siren_x3.hip -DX3_DBG_TIMES).  Prints, for one wave of each team of
workgroup 0, the shader-clock deltas between consecutive stamps of its second tile.
usage: ISO_DEV_LIB=tools/variants/libiso_times.so python tools/siren_stage_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.sdf_models import PackedSiren, Siren  # noqa: E402

P, H, L, NW = 1000000, 256, 3, int(os.environ.get("X3_NW", "8"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Siren(hidden_size=H, n_layers=L).to(dev)
pts = torch.nn.functional.normalize(torch.randn(P, 3), dim=-1).to(dev).contiguous()
ps = PackedSiren(m, dev)
sdf = torch.empty((P,), dtype=torch.float32, device=dev)
grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
lib = _lib.load()
ws = ps.workspace(P)
for _ in range(3):
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), H, L,
              ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())
torch.cuda.synchronize()
NG = 48 // NW
stash_floats = 256 * NW * L * NG * 512          # X3Shape::kStashPerWg(L) x the 256 workgroups of the 96-point shape
tail = ws[: stash_floats * 4].view(torch.int64)[-NW * 128:].cpu().view(NW, 128)
names = ["start", "pts"] + ["skew", "L0"]
for w in range(NW) if os.environ.get("X3_ALL_WAVES") else (0, NW // 2):
    t = tail[w]
    n = int((t != 0).sum())
    d = (t[1:n] - t[:n - 1]).tolist()
    print("wave %d: %d stamps, total %d cycles" % (w, n, int(t[n - 1] - t[0])))
    print("  " + " ".join("%d" % x for x in d))
