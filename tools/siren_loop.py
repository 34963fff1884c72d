"""Keep iso_siren_sdf_grad (the bench's dominant kernel) busy for SECONDS on 1 M points; prints ms per evaluation pass.
usage: [ISO_SIREN_PS=1] [ISO_SIREN_GEMM=f32] python tools/siren_loop.py [SECONDS] [P]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib  # noqa: E402
from iso_points_amd.sdf_models import PackedSiren, Siren  # noqa: E402

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Siren(hidden_size=256, n_layers=3).to(dev)
if os.environ.get("ISO_SIREN_GEMM") == "f32":
    _lib.call("iso_siren_set_gemm_mode", 0)
g = torch.Generator().manual_seed(0)
pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev).contiguous()
ps = PackedSiren(m, dev)
sdf = torch.empty((P,), dtype=torch.float32, device=dev)
grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
ws = ps.workspace(P)


def run():
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), ps.hidden,
              ps.n_hidden, ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.time()
n = 0
first = last = None
while time.time() - t0 < SEC:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    first = first if first is not None else ms
    last = ms
    n += 20
print("siren_loop PS=%s GEMM=%s: %d passes of %d evaluations in %.2f s; ms per pass: first block %.3f, last block %.3f"
      % (os.environ.get("ISO_SIREN_PS", "0"), os.environ.get("ISO_SIREN_GEMM", "split16"), n, P, time.time() - t0, first, last))
