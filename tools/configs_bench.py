"""One number per BASELINE.json config, single MI355X (SURVEY 8(d) input definitions).
cfg 3 is bench.py's own line (not repeated here).  Writes gpurun_out/configs_bench.json.
usage: python tools/configs_bench.py"""
import json, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_common import timeit
from oracle import iso_oracle as O   # model definitions only
from util import fitted_siren
from iso_points_amd.cameras import look_at_view, perspective
from iso_points_amd.dist import IsoCycle, sphere_silhouette
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from iso_points_amd.point_processing import farthest_sampling
from iso_points_amd.rasterizer import PointsRasterizationSettings
from iso_points_amd.sdf_models import SphereSDF

dev = torch.device("cuda:0")
res = {}


def sphere_cloud(P, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    return (p + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)


def put(name, **kw):
    res[name] = kw
    print(name, kw, flush=True)


# cfg 1: 10 k points in the cube, analytic unit sphere, one Newton step
torch.manual_seed(0)
pts = ((torch.rand(1, 10000, 3) - 0.5) * 2).to(dev)
sph = SphereSDF().to(dev)
proj = UniformProjection(proj_max_iters=1)
r = proj._project_points(sph, pts, full_lengths(pts), proj_max_iters=1)
t = timeit(lambda: proj._project_points(sph, pts, full_lengths(pts), proj_max_iters=1), warm=3, rep=20)
put("cfg1_sphere_10k_T1", ms=t, Mpoints_s=1e4 / t / 1e3, converged=float(r.mask.float().mean()),
    note="launch-latency bound: 10 k points are 40 workgroups")

# cfg 2: 100 k points, SIREN 4x256 fitted, project(T=10) + resample(sample_iters=1, knn_k=8)
siren = fitted_siren(O, 256, 3, seed=0, fit=200).to(dev)
pts = sphere_cloud(100000, 2)
num = full_lengths(pts)
up = UniformProjection(proj_max_iters=10, knn_k=8, sample_iters=1)


def cfg2():
    r0 = up._project_points(siren, pts, num, proj_max_iters=10)
    return up.resample(siren, r0.points, r0.normals, num, sample_iters=1)


r = cfg2()
t = timeit(cfg2, warm=2, rep=10)
put("cfg2_siren_100k_project_resample", ms=t, Mpoints_s=1e5 / t / 1e3, converged=float(r.mask.float().mean()))

# cfg 4: 4 M points, IDR 8x512 (geometric init), projection T=10: whole cloud on one GPU and the 1/8 share
torch.manual_seed(0)
idr = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6).to(dev)
# (the strong-scaling showcase: the projection is per-point work with no exchange; world N = the slowest of the N x-slab
#  shards of the SAME 4 M cloud, each projected alone on this GPU -- the per-rank compute of an N-GPU run)
from iso_points_amd.dist import slab_order, shard_bounds
g = torch.Generator().manual_seed(4)
P4 = 4000000
x4 = (torch.nn.functional.normalize(torch.randn(1, P4, 3, generator=g), dim=-1) * 0.6 +
      0.03 * (torch.rand(1, P4, 3, generator=g) - 0.5)).to(dev)
pr = UniformProjection(proj_max_iters=10)
pr.reuse_packed = True
t_whole = None
for world in (1, 2, 4, 8):
    order = slab_order(x4[0], world)
    xs = x4[:, order].contiguous()
    worst, conv = 0.0, 1.0
    for rank in (range(world) if world <= 2 else (0, world // 2, world - 1)):       # first, middle and last slab
        lo, hi = shard_bounds(P4, world, rank)
        x = xs[:, lo:hi].contiguous()
        n = full_lengths(x)
        r = pr._project_points(idr, x, n, proj_max_iters=10)
        t = timeit(lambda: pr._project_points(idr, x, n, proj_max_iters=10), warm=1, rep=3)
        worst, conv = max(worst, t), min(conv, float(r.mask.float().mean()))
        del x
    t_whole = worst if world == 1 else t_whole
    put("cfg4_idr8x512_project_T10_world%d" % world, points_per_rank=P4 // world, slowest_shard_ms=worst,
        Mpoints_s=P4 / worst / 1e3, compute_ceiling_x=t_whole / worst, converged=conv)
del x4, xs

# cfg 5: 500 k iso-points, loss-weighted insert around 5 000 FPS reference points, splat fwd/bwd at the
# reference's largest square images (it cannot do 1200 x 1600: square only, <= 1344)
P = 500000
pts = torch.nn.functional.normalize(sphere_cloud(P, 5), dim=-1)
num = full_lengths(pts)
t_fps = timeit(lambda: farthest_sampling(pts, num, 5000 / P), warm=1, rep=3)
ref_pts = farthest_sampling(pts, num, 5000 / P)[0][0]
g = torch.Generator().manual_seed(55)
metric = torch.exp(3 * torch.randn(ref_pts.shape[0], 1, generator=g)).to(dev)


class Ref(object):
    def points_packed(s): return ref_pts
    def features_packed(s): return metric
    def num_points_per_cloud(s): return torch.tensor([ref_pts.shape[0]], device=dev)
    def __len__(s): return 1


ins = UniformProjection(knn_k=8)
out = ins.insert(Ref(), pts, num)
t_ins = timeit(lambda: ins.insert(Ref(), pts, num), warm=1, rep=5)
put("cfg5_fps_5000_of_500k", ms=t_fps)
put("cfg5_insert_500k", ms=t_ins, children=int(out[3].sum()))
from iso_points_amd.rasterizer import SurfaceSplatting, _C, _visible_and_radius, composite, image_hw
for S in (1024, 1344, (1200, 1600)):      # the last: configs[4]'s own frame (H != W is beyond the reference)
    rs = PointsRasterizationSettings(image_size=S, points_per_pixel=8, cutoff_threshold=1.0, depth_merging_threshold=0.05,
                                     radii_backward_scaler=10, backface_culling=True, Vrk_isotropic=True, bin_size=None)
    views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(1)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    ss = SurfaceSplatting(raster_settings=rs)
    cloud = pts[0].contiguous()
    H, W = image_hw(S)

    def splat():
        fr = ss.front(cloud, cloud, views, projs, features_from_normals=True)
        idx, zb, qv, occ, img = _C.splat_points(fr["ndc"], fr["ellipse_params"], fr["cutoff_threshold"], fr["radii"],
                                                fr["first_idx"], fr["num_points"], 0.05, S, 8, max_pts=P,
                                                pair_capacity=8 * P, composite_with=(fr["scaler"], fr["features"], True, 1e-4))
        alpha = img[..., 3]
        occ_grad = 2.0 * (alpha - 0.5) / alpha.numel()
        zg = torch.zeros_like(zb)
        zg[..., 0] = 1e-3 / alpha.numel()
        vis, rs_ = _visible_and_radius(idx, fr["radii"], fr["first_idx"], fr["num_points"], 10.0, max_pts=P)
        return _C._backward(fr["ndc"], fr["radii"], occ_grad, fr["first_idx"], fr["num_points"], visible=vis, rs=rs_,
                            idx=idx, grad_zbuf=zg, max_pts=P)

    splat()
    t = timeit(splat, warm=1, rep=5)
    put("cfg5_splat_fwd_bwd_500k_%dx%d_1view" % (H, W), ms=t, Mpoints_s=P / t / 1e3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "configs_bench.json"), "w"), indent=1)
