"""What ONE rank of an N-GPU run of the bench cycle computes, timed on one GPU: the collectives
are replaced by copies from a cached single-GPU run (all_gather_rows) or skipped (all_reduce), so
the figure is rank 0's compute + launch time per step without any wire time -- the ceiling of the
strong-scaling curve the driver measures, and a per-stage view of what does not shrink with N.
usage: python tools/rank_share_bench.py [--model sphere|siren]"""
import argparse, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd.dist import Comm, shard_bounds
from iso_points_amd.sdf_models import SphereSDF

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="siren")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--world", type=int, default=0, help="only this N (for kernel traces); default 1,2,4,8")
args = ap.parse_args()
dev = torch.device("cuda:0")


class ShareOfN(Comm):
    """rank 0 of `world`; gathers are filled from the tensors a single-GPU run produced."""

    def __init__(self, world, full):
        super().__init__(enabled=False)
        self.world, self.rank, self.full, self.k = world, 0, full, 0

    def all_gather_rows(self, x_local, n_total):
        if self.world == 1:
            return x_local
        ref = self.full[self.k % len(self.full)]
        self.k += 1
        out = ref.clone()
        lo, hi = shard_bounds(n_total, self.world, 0)
        out[lo:hi] = x_local
        return out

    def all_reduce_(self, x, op="sum"):
        return x


if args.model == "siren":
    model = bench.fitted_siren(dev)             # the bench's own network
else:
    model = SphereSDF().to(dev)


def stage_times(cyc, comm, steps):
    """per-stage HIP-event times of IsoCycle.step: the four stage methods wrapped with events."""
    c = cyc.cyc
    names = ["project_resample", "splat_forward", "composite_band", "backward"]
    log = []

    def wrap(name):
        fn = getattr(c, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            log.append((name, e0, e1))
            return r
        setattr(c, name, timed)
        return fn
    saved = {n: wrap(n) for n in names}
    tot = 0.0
    for _ in range(steps):
        comm.k = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c.step(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) / steps
    for n, fn in saved.items():
        setattr(c, n, fn)
    acc = dict.fromkeys(names, 0.0)
    for n, a, b_ in log:
        acc[n] += a.elapsed_time(b_) / steps
    acc["glue (gathers, features, loss gradient)"] = tot - sum(acc.values())
    return acc


# the single-GPU run whose intermediate clouds stand in for the other ranks' rows
one = bench.Cycle(dev, model, Comm(enabled=False))
r0 = one.cyc._project(one.cyc.pts0_local, 10)
r1 = one.cyc.project_resample()
full = [r0.points[0].clone(), r0.normals[0].clone(), r1.points[0].clone(), r1.normals[0].clone()]
res = {}
for world in ((args.world,) if args.world else (1, 2, 4, 8)):
    comm = ShareOfN(world, full)
    cyc = bench.Cycle(dev, model, comm)
    for _ in range(2):
        comm.k = 0; cyc.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        comm.k = 0; cyc.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    comm.k = 0
    try:
        st = stage_times(cyc, comm, 5)
    except Exception as e:                      # the stage split follows IsoCycle.step; keep the total if it drifts
        st = {"error": repr(e)}
    res[world] = {"ms_per_step_rank0": ms, "ceiling_speedup": None, "stages_ms": st}
    print(world, "ms/step %.3f" % ms, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
if 1 in res:
    for w in res:
        res[w]["ceiling_speedup"] = res[1]["ms_per_step_rank0"] / res[w]["ms_per_step_rank0"]
    print({w: round(res[w]["ceiling_speedup"], 2) for w in res})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rank_share_%s.json" % args.model), "w"), indent=1)
