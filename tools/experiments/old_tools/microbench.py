"""Per-stage timings of the hot path on one GPU (not the judged bench; see bench.py).
usage: python tools/microbench.py [P] [--siren]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import frnn, _lib  # noqa: E402
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths  # noqa: E402
from iso_points_amd.sdf_models import SphereSDF, Siren  # noqa: E402


if os.environ.get("ISO_DEV_LIB"):
    from iso_points_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])

def timeit(fn, warm=2, rep=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(rep):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    pts = (pts + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)
    num = full_lengths(pts)
    proj = UniformProjection(knn_k=8)
    sphere = SphereSDF().to(dev)
    res = {}
    res["project_sphere_T10"] = timeit(lambda: proj._project_points(sphere, pts, num, proj_max_iters=10))
    r0 = proj._project_points(sphere, pts, num, proj_max_iters=10)
    p0, n0 = r0.points, r0.normals
    diag = (p0.max(dim=1).values - p0.min(dim=1).values).norm(dim=-1)
    radius = (torch.sqrt(diag / num.float()) * 8).contiguous()
    res["frnn_build"] = timeit(lambda: frnn.build_grid(p0, num, radius))
    grid = frnn.build_grid(p0, num, radius)
    res["frnn_query_K9_nn"] = timeit(lambda: frnn.frnn_grid_points(p0, p0, num, num, K=9, r=radius, grid=grid, return_nn=True))
    res["frnn_query_K9"] = timeit(lambda: frnn.frnn_grid_points(p0, p0, num, num, K=9, r=radius, grid=grid, return_nn=False))
    res["frnn_query_K7_r0.2"] = timeit(lambda: frnn.frnn_grid_points(p0, p0, num, num, K=7, r=0.2, grid=grid, return_nn=False))
    _, idxs, _, _ = frnn.frnn_grid_points(p0, p0, num, num, K=9, r=radius, grid=grid)
    idx = idxs[..., 1:]
    inv_sigma = (num.float() / diag).reshape(1).contiguous()
    res["repulse"] = timeit(lambda: proj.repulsion_step(p0, n0, idx, inv_sigma))
    res["resample_sphere"] = timeit(lambda: proj.resample(sphere, p0, n0, num, sample_iters=1))
    if "--siren" in sys.argv:
        torch.manual_seed(0)
        m = Siren(hidden_size=256, n_layers=3).to(dev)
        from iso_points_amd.sdf_models import siren_sdf_and_grad
        res["siren_eval_1"] = timeit(lambda: siren_sdf_and_grad(m, pts[0]), warm=1, rep=5)
        res["project_siren_T10_random_weights"] = timeit(lambda: proj._project_points(m, pts, num, proj_max_iters=10), warm=1, rep=3)
        flops = 2 * 2 * (3 * 256 * 256) * P  # hidden layers fwd+bwd
        print("siren eval: %.2f TFLOP/s (hidden-layer MACs only)" % (flops / res["siren_eval_1"] / 1e9))
    for k, v in res.items():
        print("%-36s %9.3f ms   %8.1f Mpts/s" % (k, v, P / v / 1e3))


if __name__ == "__main__":
    main()
