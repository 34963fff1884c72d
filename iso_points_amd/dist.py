"""The iso-point cycle on one or several MI355X (one process per GPU, torch.distributed:
backend "nccl" = RCCL over xGMI; "gloo" works too and is what the CPU / one-GPU tests use).

The reference has no distributed layer; this is the design SURVEY 8(e) / the north star
ask for.  The cloud is sharded by SPACE: the ranks hold consecutive x-slabs (balanced by point count;
`slab_order`), i.e. contiguous ranges of the brick-sorted order of csrc/bricks.hip, and the global
point order is rank-major, so every index the single-GPU cycle produces keeps its meaning.

  stage                          partition                        exchange (per cycle)
  -----------------------------  -------------------------------  -------------------------------------------
  Newton projection (T=10, T=3)  own points                       none
  FRNN tree + repulsion          own points; neighbours = own     all-gather of 8-float bounding boxes, then ONE
   (fused, csrc/bricks.hip)      bricks + imported halo cells     all-gather of the halo cells (points + normals +
                                                                  ids of the points within one fine cell of
                                                                  another slab: 32 B each)
  renderable mask, K=7           own points; halo as above        boxes + per-view counts (64 B), halo cells
   bandwidth h (fused)                                            (points + view masks)
  filter / compaction / EWA      own points -> own packed rows    none
   set-up (fused front end)
  tile binning, raster,          tiles: a band of 16-px tile      ONE all-to-all of the packed rows, each row only
   compositing, loss             rows per rank                    to the band(s) its box touches (52 B per record,
                                                                  csrc/band.hip); the band works on local row ids
  backward z (pixel-major,       own band -> per-record sums      all-reduce(max) of the scale (8 B); the reverse
   fixed point) + visible flags                                   all-to-all returns 16 B per record to its owner
  median radius                  own visible rows                 3 all-reduce(sum) of one pass's histograms (8 KB
                                                                  per view) of the radix select
  backward xy (point-major)      own rows                         all-reduce(sum) of the occ_grad bands (4 B/pixel)

Nothing is replicated: binning, raster, compositing and the z scatter see a band's rows only, everything per
point stays with the point's owner.  No collective sits inside a kernel, none
needs a size from the host: buffers have calibrated capacities and device-side counts (`calibrate`,
`check`).  With world == 1 every exchange disappears and the class is the single-GPU cycle bench.py
times.  The cycle is written as a generator that yields its exchanges, so the same code runs under a
process group (`step`) and in lock-step inside one process (`run_lockstep`: tests, and the per-rank
compute share of tools/rank_share_bench.py).
"""
import math

import torch

from . import _lib
from . import bricks
from .levelset_sampling import UniformProjection, full_lengths
from .rasterizer import (PointFragments, PointsRasterizationSettings, SurfaceSplatting, _C, _f32c,
                         _visible_and_radius, median_radius)


# ----------------------------------------------------------------------------- pure helpers
def shard_bounds(n, world, rank):
    """Balanced contiguous split of range(n): the first n % world ranks get one extra item."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_shard_bounds(n, world):
    return [shard_bounds(n, world, r) for r in range(world)]


def _cell_key(points, bits=10):
    """30-bit Morton key of a (P,3) cloud on a 2^bits grid over its bounding box (z-order curve: consecutive
    keys are neighbouring cells)."""
    mn = points.min(0).values
    ext = (points.max(0).values - mn).clamp_min(1e-20)
    q = ((points - mn) / ext * float(2 ** bits - 1)).to(torch.int64).clamp_(0, 2 ** bits - 1)

    def spread(v):                                   # 10 bits -> every third bit of 30
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def slab_order(points, world, local="cell"):
    """Permutation that puts a (P,3) cloud into the order the job works in: x-slab major (stable sort by x; rank r
    of `world` then holds rows shard_bounds(P, world, r) of the permuted cloud) and, inside a slab, along the
    z-order curve of a 1024^3 grid (`local="cell"`: consecutive rows are neighbours in space, so the atomics of the
    grid build, the gathers of the raster and the per-point stores of the fused kernels stay within a few cache
    lines; `local=None`: rows of a slab stay in x order).  One sort at set-up, not on the per-cycle path; every
    result of the cycle is per point, so the order only permutes them."""
    P = points.shape[0]
    if world == 1 and local is None:
        return torch.arange(P, device=points.device)
    by_x = torch.sort(points[:, 0], stable=True).indices
    if local is None:
        return by_x
    slab = torch.empty(P, dtype=torch.int64, device=points.device)
    for r in range(world):
        lo, hi = shard_bounds(P, world, r)
        slab[by_x[lo:hi]] = r
    key = (slab << 30) | _cell_key(points)
    return torch.sort(key, stable=True).indices


class Comm(object):
    """Thin wrapper over a torch.distributed process group (None = single process)."""

    def __init__(self, group=None, enabled=None):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.bytes_log = []          # (kind, bytes contributed by this rank) of the last cycle

        self.on_mark = None          # callback(name, arg) for the ("mark", name, arg) requests (bench timing)

    def execute(self, req):
        """One request of the cycle generator: ("all_gather", x[, out]) -> (world, *x.shape) tensor,
        ("all_to_all", x (world, ...)[, out]) -> (world, ...) tensor (equal splits: segment d of x goes to rank d),
        ("all_reduce", x, "sum"|"max") -> x reduced in place, ("mark", name, arg) -> None."""
        kind, x = req[0], req[1]
        if kind == "mark":
            if self.on_mark is not None:
                self.on_mark(x, req[2])
            return None
        self.bytes_log.append((kind, x.numel() * x.element_size()))
        if kind == "all_gather":
            out = req[2] if len(req) > 2 and req[2] is not None else x.new_empty((self.world,) + tuple(x.shape))
            self.dist.all_gather_into_tensor(out.view(-1), x.contiguous().view(-1), group=self.group)
            return out
        if kind == "all_to_all":          # x: (world, ...) segment d goes to rank d; out[s] = what rank s sent here
            out = req[2] if len(req) > 2 and req[2] is not None else torch.empty_like(x)
            if x.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # gloo moves host memory only (the CPU / one-GPU test configurations; RCCL takes the device buffers)
                hx = x.contiguous().view(-1).cpu()
                ho = torch.empty_like(hx)
                self.dist.all_to_all_single(ho, hx, group=self.group)
                out.view(-1).copy_(ho)
            else:
                self.dist.all_to_all_single(out.view(-1), x.contiguous().view(-1), group=self.group)
            return out
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX}
        self.dist.all_reduce(x, op=ops[req[2]], group=self.group)
        return x

    # host-side agreement (set-up / calibration only)
    def max_int(self, v, device):
        if self.world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


class _Single(object):
    world, rank = 1, 0
    bytes_log = []
    on_mark = None

    def execute(self, req):
        assert req[0] == "mark", "a single rank has nothing to exchange"
        if self.on_mark is not None:
            self.on_mark(req[1], req[2])
        return None

    def max_int(self, v, device):
        return int(v)


# ----------------------------------------------------------------------------- the cycle
class IsoCycle(object):
    """project(T=10) -> resample(sample_iters=1) -> splat forward -> compositing -> backward
    (SURVEY 8(d) 'cycle') on `world` GPUs.  `points0` (1,P,3) is the WHOLE initial cloud in the order
    the job works in (for world > 1: x-slab order, see `slab_order`), identical on every rank; rank r
    keeps rows shard_bounds(P, world, r)."""

    def __init__(self, model, points0, views, projs, raster_settings=None, knn_k=8, comm=None, target=None,
                 world=None, rank=None):
        self.comm = comm or _Single()
        self.world = self.comm.world if world is None else int(world)
        self.rank = self.comm.rank if rank is None else int(rank)
        self.model = model
        self.dev = points0.device
        self.P = points0.shape[1]
        self.lo, self.hi = shard_bounds(self.P, self.world, self.rank)
        self.n_own = self.hi - self.lo
        self.pts0_local = points0[:, self.lo:self.hi].contiguous()
        self.num_local = full_lengths(self.pts0_local)
        self.knn_k = knn_k
        self.proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=knn_k, sample_iters=1)
        self.rs = raster_settings or PointsRasterizationSettings(image_size=512, points_per_pixel=8)
        self.splat = SurfaceSplatting(raster_settings=self.rs)
        self.views, self.projs = _f32c(views), _f32c(projs)
        self.N = self.views.shape[0]
        self.target = target
        self.marks = False            # yield ("mark", ...) requests around the projections (bench.py timing)
        self.use_graphs = False       # replay the segments between two exchanges as HIP graphs
        self._segs = None
        self._pool = None
        # capacities (rows / records); `calibrate` shrinks them to what the workload needs
        w = self.world
        own_max = (self.P + w - 1) // w                                   # the same on every rank (buffers are gathered)
        self.halo_cap = 0 if w == 1 else max(4096, own_max)               # worst case: every own point is exported
        self.import_cap = 0 if w == 1 else max(4096, self.P - self.P // w)  # worst case: everybody else's points
        self.rec_cap = max(self.N * own_max, 1)
        self.pair_cap = max(1 << 16, 6 * self.N * self.P // w)
        self.seg_cap = 0 if w == 1 else self.rec_cap          # records per ordered pair of ranks in the band exchange
                                                              # (worst case: every own row needed by one band)
        self.halo_cells = 2           # exchanged band of the resample grid, in fine cells (2 x 0.8 r covers the radius r)
        self.halo_cells_h = 4         # ... of the bandwidth grid: its K = 7 search has no useful radius bound (r = 0.2);
                                      # a tail query that needs more than the band is counted (check: halo_uncertified)
        self._alloc()
        self._ovf = []

    def _alloc(self):
        dev, w = self.dev, self.world
        self._segs = None             # captured graphs hold the old buffers
        self.splat._row_overflow = None   # (the front end's sticky row-capacity flag: a fresh one with the new buffers)
        self.grid = bricks.BrickGrid(self.n_own, dev, import_max=self.import_cap)
        if w > 1:
            self.exp_buf = torch.zeros((2 * (self.halo_cap + 1) * 4,), dtype=torch.float32, device=dev)
            self.imp0 = torch.empty((self.import_cap, 4), dtype=torch.float32, device=dev)
            self.imp1 = torch.empty((self.import_cap, 4), dtype=torch.float32, device=dev)
            self.imp_count = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.wire = torch.empty((12 * self.rec_cap,), dtype=torch.float32, device=dev)
        if w == 1:
            from .rasterizer import median_radius_workspace
            self.med_ws1 = median_radius_workspace(self.N, dev)     # this cycle's own (graphs capture it; zero on entry / exit)
        if w > 1:
            lib, N = _lib.load(), self.N
            i64 = dict(dtype=torch.int64, device=dev)
            f32 = dict(dtype=torch.float32, device=dev)
            self.seg_floats = lib.iso_splat_band_segment_floats(self.seg_cap)
            self.send = torch.zeros((w, self.seg_floats), **f32)
            self.recv = torch.zeros((w, self.seg_floats), **f32)
            self.sent_row = torch.zeros((w * self.seg_cap,), dtype=torch.int32, device=dev)
            self.acc_own = torch.zeros((self.rec_cap,), **i64)
            self.vis_own = torch.zeros((self.rec_cap,), dtype=torch.uint8, device=dev)
            self.lay = torch.zeros((3, 8), **i64)                         # gid_first, first_global, num_global
            self.band_flags = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.exp_ws = torch.zeros((lib.iso_splat_band_export_workspace_bytes(self.n_own, N, w),), dtype=torch.uint8,
                                      device=dev)
            self.cap_local = max(1, min(w * self.seg_cap, N * self.P))
            cl = self.cap_local
            self.loc = {"ndc": torch.zeros((cl, 3), **f32), "ellipse_params": torch.zeros((cl, 3), **f32),
                        "cutoff_threshold": torch.zeros((cl,), **f32), "radii": torch.zeros((cl, 2), **f32),
                        "scaler": torch.zeros((cl,), **f32), "features": torch.zeros((cl, 3), **f32)}
            self.gid = torch.zeros((cl,), dtype=torch.int32, device=dev)
            self.origin = torch.zeros((cl,), dtype=torch.int32, device=dev)
            self.acc_l = torch.zeros((cl,), **i64)
            self.vis_l = torch.zeros((cl,), dtype=torch.uint8, device=dev)
            self.lay_l = torch.zeros((2, 8), **i64)                       # first_local, num_local
            self.place = torch.zeros((3 * w * N,), dtype=torch.int32, device=dev)
            self.ret = torch.zeros((w, self.seg_cap, 2), **i64)
            self.back = torch.zeros((w, self.seg_cap, 2), **i64)
            # the fragment arrays of the band: full-size (N,H,W,K) as the API has them, allocated and filled ONCE -- a
            # cycle only rewrites the band's rows, the rest stays at the "no fragment" values
            (S, W), K = self._image_hw(), int(self.rs.points_per_pixel)
            self.frag = (torch.full((N, S, W, K), -1, dtype=torch.int32, device=dev), torch.full((N, S, W, K), -1.0, **f32),
                         torch.full((N, S, W, K), -1.0, **f32), torch.zeros((N, S, W), **f32))
            self.idx_g = torch.full((N, S, W, K), -1, dtype=torch.int32, device=dev)
            self.img_out = torch.zeros((N, S, W, 4), **f32)
            self.med_ws = torch.zeros((lib.iso_splat_median_radius_workspace_bytes(N),), dtype=torch.uint8, device=dev)

    # -- stage 1/2: projection + resample -------------------------------------------------
    def _project(self, pts_local, T, follow=None):
        """Newton projection of the own points, bracketed by marks (bench.py times the SDF kernel
        between them; in graph mode they are segment boundaries).  follow: bricks.Follow, the side work of the launch
        for the stage that consumes its points (follow.done says whether the model's route did it)."""
        if self.marks:
            yield ("mark", "project_begin", T)
        r = self.proj._project_points(self.model, pts_local, self.num_local, proj_max_iters=T, follow=follow)
        if self.marks:
            yield ("mark", "project_end", T)
        return r

    def _halo_build(self, pts, nrm, payload, box, boxes, radius, knn_k, cell_scale, halo):
        """N ranks: common grid from the reduced box, halo export -> all-gather -> import, build."""
        p, lib_call, g = _lib.ptr, _lib.call, self.grid
        lib_call("iso_bricks_params", p(boxes), self.world, self.P, self.n_own, self.lo, float(radius), int(knn_k),
                 float(cell_scale), p(g.ws), g.n_max, _lib.stream())      # header for the union of the ranks' boxes
        lib_call("iso_halo_export", p(g.ws), p(pts), p(nrm), p(payload), self.n_own, p(boxes), self.world, self.rank,
                 halo, p(self.exp_buf), self.halo_cap, _lib.stream())
        gathered = yield ("all_gather", self.exp_buf)
        lib_call("iso_halo_import", p(g.ws), g.n_max, p(gathered), p(boxes), self.world, self.rank, halo,
                 self.halo_cap,
                 p(self.imp0), p(self.imp1), p(self.imp_count), self.import_cap, _lib.stream())
        g.build(pts, nrm, payload=payload, params_done=True, id_base=self.lo, n_total=self.P,
                imports=(self.imp0, self.imp1, self.imp_count))

    def _resample(self, pts, nrm, pending=False):
        """FRNN K+1 query + tangent-plane repulsion of the own points (levelset_sampling.py:254-284).  pending: the
        projection that produced pts left their bounding box in the grid's workspace (bricks.Follow)."""
        cs = bricks.RESAMPLE_CELL * self.knn_k
        if self.world == 1:
            if pending:
                self.grid.build(pts, nrm, knn_k=self.knn_k, cell_scale=cs, pending=True)
            else:
                self.grid.build(pts, nrm, knn_k=self.knn_k, cell_scale=cs)
        else:
            box = bricks.points_bbox(pts)
            boxes = yield ("all_gather", box)
            yield from self._halo_build(pts, nrm, None, box, boxes, -1.0, self.knn_k, cs, self.halo_cells)
        moved, _, _ = bricks.resample_fused(self.grid, self.knn_k + 1)
        return moved

    def project_resample(self, front_follows=False):
        """Generator: stages 1 and 2 -> ProjectionResult of the own points.  front_follows: the caller runs `_front` on the
        result next (cycle() does): only then does the second projection leave its bounding box / renderable mask behind
        for that build -- a stand-alone call must not leave a pending box in the grid's workspace (the next projection
        with a Follow would union its box with the stale one)."""
        one = self.world == 1 and not getattr(self, "no_follow", False)      # (no_follow: A/B and tests)
        ss = self.splat
        self._drop_stale_box()
        # one rank: the projections leave the bounding box of what they write in the grid's workspace (the next build
        # makes its header from it); the second one also takes the renderable mask (bricks.Follow)
        f0 = bricks.Follow(self.grid, self.n_own) if one else None
        r0 = yield from self._project(self.pts0_local, 10, follow=f0)
        self._box_pending = bool(f0 is not None and f0.done)
        moved = yield from self._resample(r0.points[0].contiguous(), r0.normals[0].contiguous(),
                                          pending=self._box_pending)
        self._box_pending = False                                            # consumed by the build
        f1 = bricks.Follow(self.grid, self.n_own, views=self.views, znear=ss.znear, zfar=ss.zfar,
                           backface_culling=self.rs.backface_culling) if (one and front_follows) else None
        r1 = yield from self._project(moved.view(1, -1, 3), 3, follow=f1)
        self._follow1 = f1 if (f1 is not None and f1.done) else None
        self._box_pending = self._follow1 is not None
        return r1

    def _drop_stale_box(self):
        """A pending box nobody consumed (an exception between a Follow projection and its build): taken out, so that the
        next Follow projection starts from an empty box."""
        if getattr(self, "_box_pending", False):
            bricks.box_take(self.grid)
            self._box_pending = False

    # -- stage 3: splat front end + forward ------------------------------------------------------
    def _front(self, pts, nrm):
        ss, N, w = self.splat, self.N, self.world
        f1 = getattr(self, "_follow1", None)
        self._follow1 = None
        msg = None
        if f1 is not None:                 # mask, scan and the grid's header came with the projection
            mask, cnt, scanned = f1.mask, f1.total, f1.scanned
        else:
            if w > 1:                      # the view totals straight into the message the ranks all-gather below
                msg = torch.empty((16,), dtype=torch.float32, device=pts.device)   # [box (8) | renderable points per view (8 x i32)]
            mask, cnt, scanned = bricks.view_mask_scan(pts, nrm, self.views, ss.znear, ss.zfar, self.rs.backface_culling,
                                                       total_out=msg[8:].view(torch.int32) if msg is not None else None)
        counts = None
        if w == 1:
            if f1 is not None:
                self.grid.build(pts, nrm, payload=mask, radius=float(ss.frnn_radius), cell_scale=bricks.H_CELL_SCALE,
                                pending=True, follow=f1)
                self._box_pending = False                                    # consumed by the build
            else:
                self.grid.build(pts, nrm, payload=mask, radius=float(ss.frnn_radius), cell_scale=bricks.H_CELL_SCALE)
            view_total = cnt
        else:
            bricks.points_bbox(pts, out=msg[:8])
            box = msg[:8]
            got = yield ("all_gather", msg)
            boxes = got[:, :8].contiguous()
            counts = got[:, 8:].contiguous().view(torch.int32)                     # (world, 8)
            view_total = counts.sum(dim=0, dtype=torch.int32)
            yield from self._halo_build(pts, nrm, mask, box, boxes, float(ss.frnn_radius), 0, bricks.H_CELL_SCALE,
                                        self.halo_cells_h)
        h = bricks.splat_h_fused(self.grid, mask, view_total, N)
        fr = ss.front_setup(pts, nrm, self.views, self.projs, mask, h, features_from_normals=True, out=self.wire,
                            capacity=self.rec_cap, scanned=scanned)
        if w == 1:
            fr["own_first"], fr["own_num"], fr["max_pts"], fr["rows"] = fr["first_idx"], fr["num_points"], self.P, self.rec_cap
            fr["local_first"] = fr["first_idx"]
            return fr
        # the band exchange (csrc/band.hip): every own row goes to the ranks whose band of tile rows it touches
        p, lib_call = _lib.ptr, _lib.call
        S, W = self._image_hw()
        lib_call("iso_splat_band_export", p(fr["ndc"]), p(fr["ellipse_params"]), p(fr["radii"]), p(fr["scaler"]),
                 p(fr["features"]), p(fr["first_idx"]), p(fr["num_points"]), N, self.n_own, p(counts), w, self.rank, S, W,
                 self.seg_cap, p(self.send), p(self.sent_row), p(self.acc_own), p(self.vis_own), p(self.lay[0]),
                 p(self.lay[1]), p(self.lay[2]), p(self.band_flags), p(self.exp_ws), self.exp_ws.numel(), _lib.stream())
        recv = yield ("all_to_all", self.send, self.recv)
        L = self.loc
        lib_call("iso_splat_band_import", p(recv), w, N, self.seg_cap, self.cap_local, float(self.rs.cutoff_threshold),
                 p(L["ndc"]), p(L["ellipse_params"]), p(L["cutoff_threshold"]), p(L["radii"]), p(L["scaler"]),
                 p(L["features"]), p(self.gid), p(self.origin), p(self.acc_l), p(self.vis_l), p(self.lay_l[0]),
                 p(self.lay_l[1]), p(self.place), p(self.band_flags), _lib.stream())
        out = dict(L)
        out.update({"first_idx": self.lay_l[0, :N], "num_points": self.lay_l[1, :N],          # the band's local rows
                    "own": fr,                                                                 # this rank's own rows
                    "own_first": self.lay[0, :N], "own_num": fr["num_points"],                 # ... inside the global layout
                    "local_first": fr["first_idx"],                                            # ... inside its own arrays
                    "first_global": self.lay[1, :N], "num_global": self.lay[2, :N],
                    "src": fr["src"], "mask": mask, "h": h, "max_pts": self.cap_local, "rows": self.cap_local,
                    "own_counts": fr["view_total"], "counts": counts})
        return out

    def _image_hw(self):
        from .rasterizer import image_hw
        return image_hw(self.rs.image_size)

    def splat_forward(self, fr):
        """Tile binning + raster of this rank's band of tile rows, the image composited in the same
        kernel (renderer.py:53-78).  Returns (fragments, image with the own band filled).  N ranks: the inputs are
        the rows the band received (local ids); fragments.idx carries the GLOBAL row ids (= the single-GPU lists),
        the local lists stay in self._idx_l for the backward pass."""
        rs = self.rs
        S, K = rs.image_size, int(rs.points_per_pixel)
        H, W = self._image_hw()
        T = _lib.load().iso_splat_tiles_per_side(H)
        self.band = shard_bounds(T, self.world, self.rank)
        self._ovf = []
        many = self.world > 1
        idx, zbuf, qv, occ, img = _C.splat_points(
            fr["ndc"], fr["ellipse_params"], fr["cutoff_threshold"], fr["radii"], fr["first_idx"], fr["num_points"],
            rs.depth_merging_threshold, S, K, 0, 0, tile_rows=self.band if many else None,
            out=self.frag if many else None, image_out=self.img_out if many else None,
            max_pts=fr["max_pts"], pair_capacity=self.pair_cap, overflow_out=self._ovf,
            composite_with=(fr["scaler"], fr["features"], True, 1e-4), tile_cnt_ws=self._tile_cnt_ws(self.N * T * _lib.load().iso_splat_tiles_per_side(W) + 1),
            mark_visible=None if many else fr.get("visible"))       # one rank: the visible flags of the backward pass
        if not many:
            fr["visible_marked"] = fr.get("visible") is not None
            return PointFragments(idx, zbuf, qv, None, occ), img
        self._idx_l = idx
        y0, y1 = self.band_rows()
        _lib.call("iso_splat_band_remap", _lib.ptr(idx), _lib.ptr(self.gid), self.N, H * W, y0 * W, (y1 - y0) * W, K,
                  _lib.ptr(self.idx_g), _lib.stream())
        return PointFragments(self.idx_g, zbuf, qv, None, occ), img

    def _tile_cnt_ws(self, n_ints):
        """This cycle's own zero-on-entry tile counters (the raster call leaves them zero; graphs capture the buffer)."""
        ws = getattr(self, "_tile_cnt", None)
        if ws is None or ws.numel() < 4 * n_ints:
            ws = self._tile_cnt = torch.zeros((4 * n_ints,), dtype=torch.uint8, device=self.dev)
        return ws

    def band_rows(self):
        """Output-image pixel rows [y0, y1) of this rank's tile-row band (the image is flipped)."""
        S = self._image_hw()[0]
        if self.world == 1:
            return 0, S
        b0, b1 = self.band
        return max(S - 16 * b1, 0), S - 16 * b0

    # -- stage 4: compositing + loss gradient + backward ------------------------------------------
    def backward(self, frags, fr, occ_grad_band, zbuf_grad_band):
        """occ_grad / zbuf_grad are valid on this rank's band (zero elsewhere).  Returns grad (rows,3):
        d loss / d (NDC x, y, z) of this rank's OWN packed rows, in the layout of the rank's own arrays: view v's rows
        are fr['local_first'][v] .. + fr['own_num'][v] (fr['own_first'] is their place in the GLOBAL layout of all ranks'
        rows, which no tensor on a rank has; for world == 1 the two coincide and cover all rows); fr['src'] maps own rows
        to own points."""
        idx = frags.idx
        N, S, _, K = idx.shape
        first, num = fr["first_idx"], fr["num_points"]
        scal = float(self.rs.radii_backward_scaler)
        if self.world == 1:
            vis, rs_ = _visible_and_radius(idx, fr["radii"], first, num, scal, max_pts=fr["max_pts"], vis=fr.get("visible"),
                                           med_ws=self.med_ws1, marked=bool(fr.get("visible_marked")))
            return _C._backward(fr["ndc"], fr["radii"], occ_grad_band, first, num, visible=vis, rs=rs_, idx=idx,
                                grad_zbuf=zbuf_grad_band, max_pts=fr["max_pts"], rows_covered=True)
        dev, p = idx.device, _lib.ptr
        y0, y1 = self.band_rows()
        own = fr["own"]
        occ_grad = yield ("all_reduce", occ_grad_band.contiguous(), "sum")
        idx_l = self._idx_l                          # the band's lists in local row ids
        zmax = torch.zeros((2,), dtype=torch.int32, device=dev)
        H, W = idx.shape[1], idx.shape[2]
        band = (y1 - y0) * W                         # the band of every view in one call (grid.y = view)
        _lib.call("iso_splat_band_marks", p(idx_l[0, y0:y1]) if band else None, p(zbuf_grad_band[0, y0:y1]) if band else None,
                  N, H * W, band, K, p(self.vis_l), p(zmax), _lib.stream())
        zmax = yield ("all_reduce", zmax, "max")
        # (a rank without tile rows still derives the exponent: the call then only runs the scale kernel)
        _lib.call("iso_splat_band_z_scatter", p(idx_l[0, y0:y1]) if band else None,
                  p(zbuf_grad_band[0, y0:y1]) if band else None, N, H * W, band, K, p(zmax), p(self.acc_l), _lib.stream())
        # what the band found, back to the owners of the records: fixed-point z sums and visible flags
        _lib.call("iso_splat_band_return", p(self.acc_l), p(self.vis_l), p(self.origin), p(fr["first_idx"]),
                  p(fr["num_points"]), N, self.cap_local, p(self.ret), _lib.stream())
        back = yield ("all_to_all", self.ret, self.back)
        _lib.call("iso_splat_band_merge", p(back), p(self.sent_row), self.world, N, self.n_own, self.seg_cap,
                  p(self.acc_own), p(self.vis_own), p(self.exp_ws), _lib.stream())
        # r = median radius of the visible rows of the WHOLE cloud: every rank counts its own rows, the histograms of a
        # pass are summed over the ranks before the next pass resolves them (3 x 8 KB per view)
        lib = _lib.load()
        mws = self.med_ws                            # zero on entry, left zero by the final pass (this cycle's own: ranks
        words = lib.iso_splat_median_pass_words(N)   # that share a process in run_lockstep must not share histograms)
        hist = mws[:12 * words].view(torch.int32)
        first_own, num_own = own["first_idx"], own["num_points"]
        for ps in range(3):
            _lib.call("iso_splat_median_pass", ps, p(own["radii"]), p(self.vis_own), p(first_own), p(num_own), N, self.n_own,
                      p(mws), mws.numel(), _lib.stream())
            yield ("all_reduce", hist[ps * words:(ps + 1) * words], "sum")
        rs_ = torch.empty((N,), dtype=torch.float32, device=dev)
        _lib.call("iso_splat_median_final", p(mws), N, scal, p(rs_), _lib.stream())
        grad = _C._backward(own["ndc"], own["radii"], occ_grad, first_own, num_own, visible=self.vis_own, rs=rs_,
                            max_pts=self.n_own, rows_covered=True)
        _lib.call("iso_splat_z_finish", p(self.acc_own), p(zmax), 0, grad.shape[0], p(grad), _lib.stream())
        return grad

    # -- the whole cycle -----------------------------------------------------------------------
    def cycle(self):
        """Generator over the exchanges of one cycle; returns (projection result of the own points,
        image with the own band filled, gradient of the packed rows, fragments, front-end dict)."""
        # the SDF weights are packed once per cycle (they change once per optimiser step), not per projection
        self.proj.reuse_packed, self.proj._packed_cache = True, None
        r1 = yield from self.project_resample(front_follows=True)
        fr = yield from self._front(r1.points[0].contiguous(), r1.normals[0].contiguous())
        if self.marks:
            yield ("mark", "front_end", 0)
        frags, img = self.splat_forward(fr)
        if self.marks:
            yield ("mark", "raster_end", 0)
        # loss of SURVEY 8(d) cfg 3: mean((alpha - target)^2) [+ 1e-2 mean(rgb^2): no grad to the op]
        alpha = img[..., 3]
        y0, y1 = self.band_rows()
        # d loss / d alpha = c (alpha - target), c = 2 / #pixels, as ONE pass over the image: c alpha + (-c target);
        # the constant tensors (-c target, the zbuf gradient, both zero outside the rank's band) are made once
        cgrad = self._loss_constants(alpha, frags.zbuf, y0, y1)
        if self.world == 1:
            occ_grad = torch.add(cgrad[0], alpha, alpha=cgrad[2])
        else:
            occ_grad = torch.zeros_like(alpha)          # (all-reduced IN PLACE below: a fresh one every cycle)
            torch.add(cgrad[0][:, y0:y1], alpha[:, y0:y1], alpha=cgrad[2], out=occ_grad[:, y0:y1])
        zbuf_grad = cgrad[1]
        grad = yield from self.backward(frags, fr, occ_grad, zbuf_grad)
        return r1, img, grad, frags, fr

    def _loss_constants(self, alpha, zbuf, y0, y1):
        """(-c target, zbuf gradient, c) of the cycle's loss, c = 2 / #pixels; the zbuf gradient is 1e-3 / #pixels
        on the front-most slot of the rank's band rows and 0 elsewhere.  Built on first use (the eager warm-up pass),
        constant afterwards."""
        key = (tuple(alpha.shape), tuple(zbuf.shape), y0, y1)
        if getattr(self, "_loss_key", None) != key:
            c = 2.0 / alpha.numel()
            tgt = self.target if self.target is not None else torch.zeros_like(alpha)
            zg = torch.zeros_like(zbuf)
            zg[:, y0:y1, :, 0] = 1e-3 / alpha.numel()
            self._loss_c = ((-c) * tgt).contiguous(), zg, c
            self._loss_key = key
        return self._loss_c

    def run(self, g):
        """Drive a generator of this class (cycle, project_resample, ...) with the process group."""
        try:
            req = next(g)
            while True:
                req = g.send(self.comm.execute(req))
        except StopIteration as e:
            return e.value

    def generator(self):
        return self.graph_cycle() if self.use_graphs else self.cycle()

    def step(self):
        self.comm.bytes_log = []
        if self.use_graphs and self._segs is None:
            self.run(self.cycle())            # eager warm-up (lazy kernel attributes, allocator)
            self.run(self.graph_cycle())      # capture: nothing executes yet
        return self.run(self.generator())

    def graph_cycle(self):
        """The cycle with every segment between two requests captured once as a HIP graph and replayed
        afterwards: one graph launch instead of tens of kernel launches and allocator calls per
        segment (the per-step host time is what bounds a rank once its share of the work is small).
        Same request protocol as `cycle`; the first pass only captures -- its results are undefined."""
        if self._segs is None:
            segs = []
            self._pool = torch.cuda.graph_pool_handle()
            cap_stream = torch.cuda.Stream(device=self.dev)
            g = self.cycle()
            incoming, first, final = None, True, None
            while True:
                graph = torch.cuda.CUDAGraph()
                done = False
                req = None
                try:
                    # thread-local capture mode: the process group's watchdog thread polls its events while a
                    # segment is being captured; in the default (global) mode that would invalidate the capture
                    with torch.cuda.graph(graph, pool=self._pool, stream=cap_stream, capture_error_mode="thread_local"):
                        req = next(g) if first else g.send(incoming)
                except StopIteration as e:
                    final, done = e.value, True
                first = False
                if done:
                    segs.append((graph, None, None))
                    break
                incoming = yield req
                segs.append((graph, req, incoming))
            self._segs, self._final = segs, final
            return final
        import os
        dbg = os.environ.get("ISO_GRAPH_DEBUG")
        for k, (graph, req, static) in enumerate(self._segs):
            graph.replay()
            if dbg:
                torch.cuda.synchronize()
                print("rank %d segment %d (%s) replayed" % (self.rank, k, req[0] if req else "end"), flush=True)
            if req is None:
                return self._final
            if req[0] in ("all_gather", "all_to_all"):
                got = yield (req[0], req[1], static)
                if got is not static:
                    static.copy_(got)
            else:
                yield req

    # -- capacities ------------------------------------------------------------------------------------
    def usage(self, fr=None):
        """Host read of the device-side counts of the last cycle (set-up / tests only)."""
        hdr = self.grid.header()
        c = self.grid.counters_since_last()           # summed over every grid built since the last call (all cycles)
        hdr["tail"], hdr["overflow_bricks"], hdr["tail_h"] = c[1], c[2], c[3]
        u = {"grid": hdr, "halo_export_overflow": c[4], "halo_import_overflow": c[5], "halo_uncertified": c[6],
             "pair_overflow": int(self._ovf[0].item()) if self._ovf else 0}
        if self.world > 1:
            u["halo_exported"] = int(self.exp_buf[:1].view(torch.int32).item())
            u["halo_imported"] = int(self.imp_count.item())
            u["band_overflow"] = int(self.band_flags.item())          # bit 0: a send segment, bit 1: the local arrays
            u["segment_records"] = int(self.send[:, 8].contiguous().view(torch.int32).max().item())   # largest segment wanted
            u["band_rows"] = int((self.lay_l[0, self.N - 1] + self.lay_l[1, self.N - 1]).item())       # rows this band received
        if fr is not None:
            u["own_rows"] = int(fr["own_num"].sum().item())
            own = fr.get("own", fr)
            if own.get("row_overflow") is not None:
                u["row_overflow"] = int(own["row_overflow"].item())          # the front end dropped rows at rec_cap
        return u

    def check(self, fr=None, usage=None):
        u = usage if usage is not None else self.usage(fr)
        bad = [k for k in ("halo_export_overflow", "halo_import_overflow", "halo_uncertified", "pair_overflow", "band_overflow",
                           "row_overflow")
               if u.get(k)]
        if self.world > 1 and (u["halo_exported"] > self.halo_cap or u["halo_imported"] > self.import_cap):
            bad.append("halo capacity")
        if fr is not None and self.world > 1 and u["own_rows"] > self.rec_cap:
            bad.append("row capacity")
        if bad:
            raise RuntimeError("IsoCycle: buffer overflow (%s): %s -- raise the capacities / calibrate()" % (bad, u))
        return u

    def calibrate(self, margin=1.5):
        """One synchronised cycle, then shrink the exchange buffers to `margin` x what it used (agreed
        over the ranks).  Untimed set-up; the capacities stay fixed afterwards and `check` reports
        an overflow."""
        self.grid.counters_since_last()              # forget what earlier cycles counted
        out = self.step()
        u = self.usage(out[4])
        if self.world > 1:
            # a bandwidth query near a slab face whose 7th neighbour lies beyond the exchanged band: widen it
            for _ in range(6):
                if self.comm.max_int(u["halo_uncertified"], self.dev) == 0:
                    break
                self.halo_cells_h *= 2
                self._segs = None                     # captured segments have the old band width baked in
                out = self.step()
                u = self.usage(out[4])
        u = self.check(out[4], usage=u)
        if self.world > 1:
            c = self.comm
            self.halo_cap = max(1024, int(margin * c.max_int(u["halo_exported"], self.dev)))
            self.import_cap = max(1024, int(margin * c.max_int(u["halo_imported"], self.dev)))
            self.rec_cap = max(1024, int(min(margin, 1.25) * c.max_int(u["own_rows"], self.dev)))
            self.seg_cap = max(1024, int(min(margin, 1.25) * c.max_int(u["segment_records"], self.dev)))
            self._alloc()
        return u


def run_lockstep(cycles, timer=None):
    """Run one cycle of every rank of a job inside ONE process (all ranks on one device): the generators
    advance in lock-step and the exchanges are done by hand.  Returns the list of the ranks' results.
    `timer(rank)` (optional) is a context manager entered around every compute segment of a rank --
    tools/rank_share_bench.py sums the device time per rank with it."""
    import contextlib
    gens = [c.generator() if hasattr(c, "generator") else c.cycle() for c in cycles]
    tm = timer or (lambda r: contextlib.nullcontext())
    reqs, results = [None] * len(gens), [None] * len(gens)
    live = list(range(len(gens)))
    send = [None] * len(gens)
    first = True
    while live:
        for r in list(live):
            try:
                with tm(r):
                    reqs[r] = next(gens[r]) if first else gens[r].send(send[r])
            except StopIteration as e:
                results[r] = e.value
                live.remove(r)
        first = False
        if not live:
            break
        assert len(live) == len(gens), "ranks left the cycle at different exchanges"
        kind = reqs[0][0]
        assert all(q[0] == kind for q in reqs)
        if kind == "mark":
            send = [None for _ in gens]
        elif kind == "all_gather":
            g = torch.stack([q[1] for q in reqs])
            send = []
            for q in reqs:
                if len(q) > 2 and q[2] is not None:
                    q[2].copy_(g)
                    send.append(q[2])
                else:
                    send.append(g.clone() if len(reqs) > 1 and getattr(cycles[0], "use_graphs", False) else g)
        elif kind == "all_to_all":
            send = []
            for r, q in enumerate(reqs):
                got = torch.stack([reqs[s_][1][r] for s_ in range(len(reqs))])
                if len(q) > 2 and q[2] is not None:
                    q[2].copy_(got)
                    got = q[2]
                send.append(got)
        else:
            st = torch.stack([q[1] for q in reqs])
            red = st.sum(dim=0) if reqs[0][2] == "sum" else st.max(dim=0).values
            send = []
            for q in reqs:
                q[1].copy_(red)
                send.append(q[1])
    return results


def sphere_silhouette(S, n_views, dist, fov_deg, device):
    """Analytic alpha target of a unit sphere centred at the origin (SURVEY 8(d) cfg 3)."""
    ax = -1 + (2 * torch.arange(S, device=device) + 1.0) / S
    rr = (1.0 / math.sqrt(dist * dist - 1.0)) / math.tan(math.radians(fov_deg) / 2)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    return ((xx ** 2 + yy ** 2) <= rr ** 2).float()[None].expand(n_views, S, S).contiguous()
