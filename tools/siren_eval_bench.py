"""Time one fused SIREN SDF+gradient evaluation (iso_siren_sdf_grad) on P points.
usage: [ISO_DEV_LIB=tools/variants/libiso_X.so] [ISO_SIREN_GEMM=f32] python tools/siren_eval_bench.py [P] [H] [L]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib  # noqa: E402

if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.sdf_models import PackedSiren, Siren  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = Siren(hidden_size=H, n_layers=L).to(dev)
    g = torch.Generator().manual_seed(0)
    pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev).contiguous()
    ps = PackedSiren(m, dev)
    sdf = torch.empty((P,), dtype=torch.float32, device=dev)
    grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
    ws = ps.workspace(P)

    def run():
        _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), ps.hidden,
                  ps.n_hidden, ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    t = ts[len(ts) // 2]
    flop = 2.0 * (2 * L * H * H + 2 * 3 * H + 2 * H) * P
    print("%-28s P=%d H=%d L=%d  %.3f ms  %.1f Mevals/s  %.1f TFLOP/s(f32-equivalent)"
          % (os.path.basename(os.environ.get("ISO_DEV_LIB", "default")), P, H, L, t, P / t / 1e3, flop / t / 1e9))


main()
