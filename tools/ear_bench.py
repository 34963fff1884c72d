"""EdgeAwareProjection.upsample (levelset_sampling.py:528-661) on 100 k iso-points of the bench's
fitted SIREN: K = 31, ratio 1.5 (5 rounds of P/10 insertions), and the candidate kernel alone
against the reference's (1,P,K,K,3) tensor statement run with torch on the same GPU.
usage: python tools/ear_bench.py [--points 100000]"""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from tools_common import timeit
from iso_points_amd import _lib, frnn
from iso_points_amd.levelset_sampling import EdgeAwareProjection, full_lengths

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=100000)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = bench.fitted_siren(dev)
P = args.points
pts = bench.sphere_cloud(P, seed=0, device=dev)
ear = EdgeAwareProjection(knn_k=31)
pts = ear._project_points(net, pts, full_lengths(pts), proj_max_iters=10).points
num = full_lengths(pts)
up, n = ear.upsample(pts, P, net, num)
t_up = timeit(lambda: ear.upsample(pts, P, net, num), warm=1, rep=5)
# the candidate statement alone
idx = ear._create_tree(pts, refresh_tree=True, num_points_per_cloud=num)
_, nrm = ear._compute_sdf_and_grad(pts, net)
nrm = torch.nn.functional.normalize(nrm, dim=-1)
knn_p, knn_n = frnn.frnn_gather(pts, idx, num).contiguous(), frnn.frnn_gather(nrm, idx, num).contiguous()
sp, cand = torch.empty(1, P, device=dev), torch.empty(1, P, 3, device=dev)


def fused():
    _lib.call("iso_ear_candidates", _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(knn_p), _lib.ptr(knn_n), P, 31, 1.0,
              _lib.ptr(sp), _lib.ptr(cand), _lib.stream())


def tensor_form():
    mid = (knn_p + 2 * pts[..., None, :]) / 3
    d = mid.unsqueeze(-2) - knn_p.unsqueeze(-3)
    edge = 2 - torch.sum(nrm.unsqueeze(-2) * knn_n, dim=-1)
    m = torch.norm(d, dim=-1) - torch.sum((d * knn_n.unsqueeze(-2)) ** 2, dim=-1)
    m = m.min(dim=-1)[0].abs().clamp_min(1e-17).sqrt()
    return (edge * m).max(dim=-1)


t_f = timeit(fused, warm=2, rep=10)
t_t = timeit(tensor_form, warm=1, rep=5)
ref = tensor_form()[0]
fused()
res = {"points": P, "K": 31, "upsample_ms": t_up, "out_points": int(n), "candidates_fused_ms": t_f,
       "candidates_tensor_form_same_gpu_ms": t_t,
       "candidates_max_rel_diff": ((sp - ref).abs().max() / ref.abs().max()).item()}
# the exact K-nearest search underneath (K + 1 = 32), cell size from the default density target
# (8 points per cell: most queries need the ring walk) and from the K-aware one knn_points uses
from iso_points_amd import point_processing as PP
r_inf = torch.full((1,), float("inf"), device=dev)
for tag, ppc in (("knn32_ms_8_per_cell", 8.0), ("knn32_ms_0.75K_per_cell", 0.75 * 32)):
    def run(ppc=ppc):
        g = frnn.build_grid(up, n, r_inf, points_per_cell=ppc)
        return frnn.frnn_grid_points(up, up, n, n, K=32, r=r_inf, grid=g, return_nn=True)
    res[tag] = timeit(run, warm=1, rep=5)
a, b = PP.knn_points(up, up, n, n, K=32, return_nn=True), None
g8 = frnn.build_grid(up, n, r_inf, points_per_cell=8.0)
d8, i8, _, _ = frnn.frnn_grid_points(up, up, n, n, K=32, r=r_inf, grid=g8)
res["knn_same_result"] = bool(torch.equal(a.idx, i8.clamp_min(0)) and torch.equal(a.dists, d8.clamp_min(0)))
# point_processing.upsample (sparsest-edge midpoints, K = 31 as UniformProjection.upsample) and wlop
res["pp_upsample_100k_to_150k_ms"] = timeit(lambda: PP.upsample(pts, int(1.5 * P), num_points=num, neighborhood_size=31), warm=1, rep=3)
res["wlop_100k_ratio0.5_ms"] = timeit(lambda: PP.wlop(pts, num, ratio=0.5), warm=1, rep=3)
print(res)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ear_bench.json"), "w"), indent=1)
