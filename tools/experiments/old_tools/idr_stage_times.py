"""Stage timeline of the split-fp16 IDR step kernel (needs -DI16_DBG_TIMES: tools/build_variant.sh times
idr_x16.hip -DI16_DBG_TIMES).  usage: ISO_DEV_LIB=tools/variants/libiso_times.so python tools/idr_stage_times.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from oracle import iso_oracle as O
from iso_points_amd.sdf_models import PackedIdr
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, NL, NB, NW = 512, 8, 2, 8
m = O.IdrSDF(hidden_size=H, n_layers=NL, skip_in=(4,), num_frequencies=6).to(dev)
P = 300000
pts = (torch.nn.functional.normalize(torch.randn(P, 3), dim=-1) * 0.6).to(dev)
pk = PackedIdr(m, dev)
sdf = torch.empty((P,), device=dev); grad = torch.empty((P, 3), device=dev)
ws = pk.workspace(P)
for _ in range(2):
    _lib.call("iso_idr_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(pk.packed), pk.hidden, pk.n_layers,
              pk.skip, pk.n_freq, 100.0, _lib.ptr(ws), ws.numel(), _lib.stream())
torch.cuda.synchronize()
NG = 2 * (H // 32 // NW) * NB
stash_floats = 256 * NW * NL * NG * 512
tail = ws[: stash_floats * 4].view(torch.int64)[-NW * 128:].cpu().view(NW, 128)
for w in (0, 4):
    t = tail[w]; n = int((t != 0).sum()); d = (t[1:n] - t[:n - 1]).tolist()
    print("wave %d: %d stamps, total %d cycles" % (w, n, int(t[n - 1] - t[0])))
    print("  " + " ".join("%d" % x for x in d))
