"""sample_uniform_iso_points (levelset_sampling.py:1405-1445), the reference's stand-alone sampler:
project 4n random points -> wlop -> project + upsample -> upsample(n) -> project, on the bench's
fitted SIREN, with per-stage times.  usage: python tools/sample_uniform_bench.py [--points 25000]"""
import argparse, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import iso_points_amd.levelset_sampling as L
import iso_points_amd.point_processing as PP

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=25000)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = bench.fitted_siren(dev)
stages = {}


def wrap(mod, name):
    fn = getattr(mod, name)

    def timed(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); stages[name] = stages.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(mod, name, timed)


for name in ("wlop", "upsample", "farthest_sampling"):
    wrap(PP, name)
wrap(L.UniformProjection, "project_points")
g = torch.Generator().manual_seed(0)
out = L.sample_uniform_iso_points(net, args.points, generator=g, device=dev)      # warm-up (loads, packs)
stages.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
out = L.sample_uniform_iso_points(net, args.points, generator=torch.Generator().manual_seed(0), device=dev)
torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
res = {"n_points": args.points, "out_points": out.shape[1], "total_ms": total,
       "stages_ms (inclusive; wlop contains farthest_sampling, project_points contains upsample)": stages}
print(res)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sample_uniform_bench.json"), "w"), indent=1)
