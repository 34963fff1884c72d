"""BASELINE.json configs[3] as a SHARDED job: 4 M points, IDR 8 x 512, x-slab shards, project (T = 10) -> halo exchange +
fused resample -> re-projection (T = 3) = IsoCycle.project_resample on N ranks in lock-step on ONE GPU
(iso_points_amd.dist.run_lockstep: the generator code a process group drives; exchanges are device copies).  HIP events
bracket every compute segment of every rank; the slowest rank's sum is the job's per-step compute time = the ceiling of
the strong-scaling curve before wire time.  Writes gpurun_out/cfg3_sharded.json.
usage: python tools/cfg3_sharded_bench.py [P] [steps]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import iso_oracle as O   # model definition only
from rank_share_bench import Timer
from iso_points_amd.cameras import look_at_view, perspective
from iso_points_amd.dist import IsoCycle, run_lockstep, slab_order
from iso_points_amd.rasterizer import PointsRasterizationSettings

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
idr = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6).to(dev)
g = torch.Generator().manual_seed(4)
base = (torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1) * 0.6 +
        0.03 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(dev)
views = torch.stack([look_at_view(3.0, 20.0, 0.0)]).to(dev)
projs = views @ perspective(30.0).to(dev)
rs = PointsRasterizationSettings(image_size=64, points_per_pixel=8)


class Stage12(object):
    """run_lockstep drives `generator()`: here stages 1 + 2 only"""
    use_graphs = False

    def __init__(self, cyc):
        self.cyc = cyc

    def generator(self):
        return self.cyc.project_resample()


out = {"points": P, "sdf": "IDR 8 x 512, skip 4, 6 frequencies (geometric init)", "stages": "project T=10 -> halo + fused resample -> project T=3",
       "worlds": {}}
t1 = None
for world in (1, 2, 4, 8):
    pts = base[:, slab_order(base[0], world)].contiguous()
    ranks = [IsoCycle(idr, pts, views, projs, raster_settings=rs, knn_k=8, world=world, rank=r) for r in range(world)]
    for c in ranks:
        c.proj.reuse_packed = True                                    # one weight image per step, as cycle() does
    jobs = [Stage12(c) for c in ranks]
    res = run_lockstep(jobs)                                          # warm-up (+ the lazily sized buffers)
    per_step = []
    for _ in range(steps):
        tm = Timer(world)
        res = run_lockstep(jobs, timer=tm)
        per_step.append(tm.per_rank_ms())
    ms = [sorted(st[r] for st in per_step)[len(per_step) // 2] for r in range(world)]
    use = [c.usage() for c in ranks]
    for c, u in zip(ranks, use):
        c.check(usage=u)
    conv = min(float(r.mask.float().mean()) for r in res)
    if t1 is None:
        t1 = max(ms)
    halo = [(u.get("halo_exported", 0), u.get("halo_imported", 0)) for u in use]
    out["worlds"][str(world)] = {"per_rank_ms": [round(x, 2) for x in ms], "slowest_rank_ms": round(max(ms), 2),
                                 "compute_ceiling_x": round(t1 / max(ms), 2), "converged_min": round(conv, 4),
                                 "halo_records_exported_imported_max": [max(h[0] for h in halo), max(h[1] for h in halo)],
                                 "halo_bytes_contributed_per_rank_max": 32 * max(h[0] for h in halo),
                                 "tail_queries_max": max(u["grid"]["tail"] for u in use)}
    print(world, out["worlds"][str(world)], flush=True)
    del ranks, jobs, res
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cfg3_sharded.json"), "w"), indent=1)
