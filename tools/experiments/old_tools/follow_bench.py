"""iso_project_sphere vs iso_project_sphere_follow (header only / header + mask) at 1 M points: us per call (HIP events).
usage: python tools/follow_bench.py   (ISO_DEV_LIB selects a library variant)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iso_points_amd import _lib, bricks
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from iso_points_amd.sdf_models import SphereSDF
from iso_points_amd.cameras import look_at_view
dev = torch.device("cuda:0")
P = 1000000
g = torch.Generator().manual_seed(0)
p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
pts = (p + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)
m = SphereSDF().to(dev)
proj = UniformProjection()
grid = bricks.BrickGrid(P, dev)
views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(4)]).to(dev).contiguous()
lens = full_lengths(pts)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for T in (10, 3):
    t0 = timed(lambda: proj._project_points(m, pts, lens, proj_max_iters=T))
    t1 = timed(lambda: proj._project_points(m, pts, lens, proj_max_iters=T, follow=bricks.Follow(grid, P)) and bricks.box_take(grid))
    t2 = timed(lambda: proj._project_points(m, pts, lens, proj_max_iters=T, follow=bricks.Follow(grid, P, views=views)) and bricks.box_take(grid))
    print("T=%d: plain %.1f us, follow(header) %.1f us, follow(header+mask) %.1f us (eager: includes host time)" % (T, t0, t1, t2))
