"""What ONE rank of an N-GPU run of the bench cycle computes, measured on one GPU: the N ranks run in
lock-step inside this process (iso_points_amd.dist.run_lockstep: the same generator code a process group
drives; exchanges are device copies), HIP events bracket every compute segment of every rank, and the
slowest rank's sum is the per-step compute time of the job = the ceiling of the strong-scaling curve
before any wire time.  Also logs the bytes every rank contributes to each collective.

usage: python tools/rank_share_bench.py [sphere|siren] [P] [steps]"""
import contextlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iso_points_amd import _lib as _L
if os.environ.get("ISO_DEV_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
import bench as B
from iso_points_amd.dist import IsoCycle, run_lockstep, slab_order, sphere_silhouette
from iso_points_amd.rasterizer import PointsRasterizationSettings
from iso_points_amd.sdf_models import SphereSDF


class Timer(object):
    def __init__(self, world):
        self.ev = [[] for _ in range(world)]

    @contextlib.contextmanager
    def __call__(self, r):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            self.ev[r].append((a, b))

    def per_rank_ms(self):
        torch.cuda.synchronize()
        return [sum(a.elapsed_time(b) for a, b in evs) for evs in self.ev]


class TraceTimer(Timer):
    """ISO_TRACE_RANK=r: the segments of rank r are bracketed by marker kernels (torch.cuda._sleep -> spin_kernel) so that
    tools/rank_sequence.py can list what ONE rank launches per cycle from a rocprofv3 kernel trace."""
    def __init__(self, world, rank):
        super().__init__(world)
        self.rank = rank

    @contextlib.contextmanager
    def __call__(self, r):
        if r == self.rank:
            torch.cuda._sleep(2000)
        try:
            with super().__call__(r):
                yield
        finally:                                  # (the rank's last segment ends with StopIteration)
            if r == self.rank:
                torch.cuda._sleep(2000)


class LogComm(object):
    """records the requests one rank yields (bytes per collective)"""
    def __init__(self):
        self.log = []


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "sphere"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else B.P_TOTAL
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    graphs = len(sys.argv) > 4 and sys.argv[4] == "graphs"
    dev = torch.device("cuda:0")
    model = SphereSDF().to(dev) if kind == "sphere" else B.fitted_siren(dev)
    views, projs = B.cameras(dev)
    rs = PointsRasterizationSettings(image_size=B.IMAGE, points_per_pixel=B.KPIX, cutoff_threshold=1.0,
                                     depth_merging_threshold=0.05, radii_backward_scaler=10, backface_culling=True,
                                     Vrk_isotropic=True, bin_size=None)
    target = sphere_silhouette(B.IMAGE, B.VIEWS, 3.0, 30.0, dev)
    base = B.sphere_cloud(P, seed=0, device=dev)
    out = {"sdf": kind, "points": P, "steps": steps, "graphs": graphs, "worlds": {}}
    t1 = None
    for world in [int(w) for w in os.environ.get("ISO_WORLDS", "1,2,4,8").split(",")]:
        pts = base[:, slab_order(base[0], world)].contiguous()
        ranks = [IsoCycle(model, pts, views, projs, raster_settings=rs, knn_k=8, target=target, world=world, rank=r)
                 for r in range(world)]
        res = run_lockstep(ranks)
        use = [c.usage(o[4]) for c, o in zip(ranks, res)]
        for _ in range(6):      # = IsoCycle.calibrate() without a process group: widen the bandwidth halo if needed
            if max(u["halo_uncertified"] for u in use) == 0:
                break
            for c in ranks:
                c.halo_cells_h *= 2
            res = run_lockstep(ranks)
            use = [c.usage(o[4]) for c, o in zip(ranks, res)]
        use = [c.check(o[4], usage=u) for c, o, u in zip(ranks, res, use)]
        if world > 1:
            for c in ranks:
                c.halo_cap = max(1024, int(1.5 * max(u["halo_exported"] for u in use)))
                c.import_cap = max(1024, int(1.5 * max(u["halo_imported"] for u in use)))
                c.rec_cap = max(1024, int(1.25 * max(u["own_rows"] for u in use)))
                c.seg_cap = max(1024, int(1.25 * max(u["segment_records"] for u in use)))
                c._alloc()
        del res
        run_lockstep(ranks)
        if graphs:
            for c in ranks:
                c.use_graphs = True
        for _ in range(3):
            run_lockstep(ranks)
        tr = os.environ.get("ISO_TRACE_RANK")
        per_step = []
        for _ in range(steps):
            tm = Timer(world) if tr is None else TraceTimer(world, min(int(tr), world - 1))
            res = run_lockstep(ranks, timer=tm)
            per_step.append(tm.per_rank_ms())
        # a rank's time = the MEDIAN of its per-step sums (one stall of the process in one of five steps -- seen once: 12 ms
        # for a 2.7 ms rank -- must not become the job's figure)
        ms = [sorted(st[r] for st in per_step)[len(per_step) // 2] for r in range(world)]
        use = [c.check(o[4]) for c, o in zip(ranks, res)]
        # bytes each rank contributes per collective: replay rank 0's generator requests
        log = []
        if world > 1:
            for c in ranks:
                c.use_graphs = False
            gens = [c.cycle() for c in ranks]
            reqs = [next(g) for g in gens]
            while True:
                log.append((reqs[0][0] + ("" if reqs[0][0] in ("all_gather", "all_to_all") else ":" + reqs[0][2]),
                            reqs[0][1].numel() * reqs[0][1].element_size()))
                if reqs[0][0] == "all_gather":
                    send = [torch.stack([q[1] for q in reqs])] * world
                elif reqs[0][0] == "all_to_all":
                    send = [torch.stack([reqs[s_][1][r_] for s_ in range(world)]) for r_ in range(world)]
                else:
                    st = torch.stack([q[1] for q in reqs])
                    red = st.sum(0) if reqs[0][2] == "sum" else st.max(0).values
                    send = [red] * world
                try:
                    reqs = [g.send(s) for g, s in zip(gens, send)]
                except StopIteration:
                    break
        if t1 is None:
            t1 = max(ms) * world if world > 1 else max(ms)
        out["worlds"][str(world)] = {
            "per_rank_ms": [round(x, 3) for x in ms], "slowest_rank_ms": round(max(ms), 3),
            "compute_ceiling_x": round(t1 / max(ms), 2),
            "collectives_bytes_per_rank": log, "bytes_per_rank_total": sum(b for _, b in log),
            "halo_exported_max": max((u.get("halo_exported", 0) for u in use), default=0),
            "halo_imported_max": max((u.get("halo_imported", 0) for u in use), default=0),
            "own_rows_max": max(u["own_rows"] for u in use),
            "segment_records_max": max((u.get("segment_records", 0) for u in use), default=0),
            "band_rows_max": max((u.get("band_rows", 0) for u in use), default=0), "halo_cells": [ranks[0].halo_cells, ranks[0].halo_cells_h],
            "grid_per_rank": [{k: u["grid"][k] for k in ("occupied", "tail", "overflow_bricks", "tail_h", "n")} for u in use]}
        print(world, out["worlds"][str(world)]["slowest_rank_ms"], out["worlds"][str(world)]["compute_ceiling_x"], file=sys.stderr)
        del ranks, res
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
