"""The iso-point cycle on one or several MI355X (one process per GPU, torch.distributed:
backend "nccl" = RCCL over xGMI; "gloo" works too and is what the CPU/1-GPU tests use).

The reference has no distributed layer; this is new design:

  stage                         partition                         exchange
  ----------------------------  --------------------------------  ---------------------------------
  Newton projection (T=10, 3)   points: contiguous slice / rank   none
  FRNN tree + repulsion         queries / moves: own slice;       all-gather of positions + normals
                                grid built over the whole cloud   (24 B/pt) -> neighbour indices are
                                                                  GLOBAL and bit-identical to 1 GPU
  splat filter / compaction     replicated (0.3 ms)               (uses the gathered cloud)
  K=7 FRNN for h                queries: slice of every view      all-reduce(sum) of h (4 B/pt-view)
  per-point EWA set-up          replicated (bit-identical)        none
  tile binning + raster         pixels: band of 16-px tile rows   none (all splats are local)
  compositing, loss gradient    own band                          all-reduce(sum) of occ_grad bands
  visible set                   own band                          all-reduce(max) of the flags
  backward xy (point-major)     points: slice of every view       none
  backward z (pixel-major)      own band                          all-reduce(sum) of z-gradients

Every collective moves O(points) or O(pixels) bytes once per cycle (about 70 MB at 1M points /
512^2 x 4 views); nothing is exchanged inside a kernel.  With world == 1 every exchange is a no-op
and the class is exactly the single-GPU cycle bench.py times.
"""
import math

import torch

from . import _lib
from . import frnn
from .levelset_sampling import UniformProjection, cloud_diag, full_lengths, with_host_lengths
from .rasterizer import (PointFragments, PointsRasterizationSettings, SurfaceSplatting, _C, _f32c,
                         _visible_and_radius, gather_with_neg_idx, median_radius)


# ----------------------------------------------------------------------------- pure helpers
def shard_bounds(n, world, rank):
    """Balanced contiguous split of range(n): the first n % world ranks get one extra item."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_shard_bounds(n, world):
    return [shard_bounds(n, world, r) for r in range(world)]


class Comm(object):
    """Thin wrapper over a torch.distributed process group (None = single process)."""

    def __init__(self, group=None, enabled=None):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() if enabled is None else enabled
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0

    def all_gather_rows(self, x_local, n_total):
        """Concatenate per-rank row blocks (shard_bounds order) into the full (n_total, ...) tensor."""
        if self.world == 1:
            return x_local
        bounds = all_shard_bounds(n_total, self.world)
        mx = max(hi - lo for lo, hi in bounds)
        tail = tuple(x_local.shape[1:])
        buf = x_local.new_zeros((mx,) + tail)
        buf[: x_local.shape[0]] = x_local
        out = x_local.new_empty((self.world * mx,) + tail)
        self.dist.all_gather_into_tensor(out, buf.contiguous(), group=self.group)
        if all(hi - lo == mx for lo, hi in bounds):
            return out
        return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(bounds)], dim=0)

    def all_reduce_(self, x, op="sum"):
        if self.world == 1:
            return x
        ops = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}
        self.dist.all_reduce(x, op=ops[op], group=self.group)
        return x


# ----------------------------------------------------------------------------- the cycle
class IsoCycle(object):
    """project(T=10) -> resample(sample_iters=1) -> splat forward -> compositing -> backward
    (SURVEY 8(d) 'cycle'), on `comm.world` GPUs.  `points0` is the WHOLE initial cloud (1,P,3) on
    this rank's device (the same on every rank); each rank works on its slice."""

    def __init__(self, model, points0, views, projs, raster_settings=None, knn_k=8, comm=None,
                 target=None):
        self.comm = comm or Comm(enabled=False)
        self.model = model
        self.P = points0.shape[1]
        self.lo, self.hi = shard_bounds(self.P, self.comm.world, self.comm.rank)
        self.pts0_local = points0[:, self.lo:self.hi].contiguous()
        self.num_local = full_lengths(self.pts0_local)
        self.proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=knn_k, sample_iters=1)
        self.rs = raster_settings or PointsRasterizationSettings(image_size=512, points_per_pixel=8)
        self.splat = SurfaceSplatting(raster_settings=self.rs)
        self.views, self.projs = _f32c(views), _f32c(projs)
        self.target = target
        self.project_hook = None      # bench.py installs timing events here

    # -- stage 1/2: projection + resample -------------------------------------------------
    def _project(self, pts_local, T):
        if self.project_hook is not None:
            return self.project_hook(lambda: self.proj._project_points(self.model, pts_local, self.num_local,
                                                                       proj_max_iters=T), T)
        return self.proj._project_points(self.model, pts_local, self.num_local, proj_max_iters=T)

    def project_resample(self):
        c, proj = self.comm, self.proj
        r0 = self._project(self.pts0_local, 10)
        pts_all = c.all_gather_rows(r0.points[0], self.P).view(1, self.P, 3)
        nrm_all = c.all_gather_rows(r0.normals[0], self.P).view(1, self.P, 3)
        num_all = full_lengths(pts_all)
        diag_all = cloud_diag(pts_all)                                  # one bounding-box pass for both uses
        diag = diag_all[0]                                              # levelset_sampling.py:254-256
        inv_sigma = (num_all.float() / diag).reshape(1).contiguous()
        if c.world == 1:
            proj._create_tree(pts_all, refresh_tree=True, num_points_per_cloud=num_all)
            idx = proj._knn_idx
        else:
            radius = (torch.sqrt(diag_all / num_all.float()) * proj.knn_k).contiguous()              # :129-131
            grid = frnn.build_grid(pts_all, num_all, radius)
            own = pts_all[:, self.lo:self.hi].contiguous()
            _, idxs, _, _ = frnn.frnn_grid_points(own, pts_all, self.num_local, num_all, K=proj.knn_k + 1,
                                                  r=radius, grid=grid)
            idx = idxs[..., 1:]
        moved = proj.repulsion_step(pts_all, nrm_all, idx, inv_sigma, first_point=self.lo)
        r1 = self._project(moved, 3)
        self.knn_idx = idx
        return r1

    # -- stage 3: splat forward ---------------------------------------------------------------
    def splat_forward(self, pts_all, nrm_all, features_all=None):
        """pts_all/nrm_all (P,3): the whole (gathered) cloud.  Returns fragments for this rank's
        band of tile rows (other pixels -1 / 0) and the filtered per-view data."""
        c, ss, rs = self.comm, self.splat, self.rs
        dev = pts_all.device
        S, K, N = int(rs.image_size), int(rs.points_per_pixel), self.views.shape[0]
        P = pts_all.shape[0]
        flags, off, lens = ss.filter_renderable(pts_all, nrm_all, self.views)
        tot = sum(lens)
        fl = [sum(lens[:i]) for i in range(N)]
        num = with_host_lengths(torch.tensor(lens, dtype=torch.int64, device=dev), lens)
        first = with_host_lengths(torch.tensor(fl, dtype=torch.int64, device=dev), fl)
        pts_f = ss.compact(pts_all, flags, off, P, tot)
        nrm_f = ss.compact(nrm_all, flags, off, P, tot)
        feat_f = ss.compact(features_all, flags, off, P, tot) if features_all is not None else None
        if c.world == 1:
            ndc, info = ss.per_point_info(pts_f, nrm_f, first, num, self.views, self.projs)
        else:
            h = self._h_share(pts_f, lens, fl, num, tot)
            c.all_reduce_(h, "sum")
            ndc, info = self._setup_with_h(pts_f, nrm_f, h, first, num)
        T = _lib.load().iso_splat_tiles_per_side(S)
        band = shard_bounds(T, c.world, c.rank)
        self.band = band
        idx, zbuf, qv, occ = _C.splat_points(ndc, info["ellipse_params"], info["cutoff_threshold"],
                                             info["radii"], first, num, rs.depth_merging_threshold, S, K,
                                             0, 0, tile_rows=band if c.world > 1 else None)
        frags = PointFragments(idx, zbuf, qv, gather_with_neg_idx(info["scaler"], idx), occ)
        filt = {"points": pts_f, "normals": nrm_f, "features": feat_f, "ndc": ndc, "num_points": num,
                "first_idx": first, **info}
        return frags, filt

    def _h_share(self, pts_f, lens, fl, num, tot):
        """This rank's part of h (rasterizer.py:367-386: K=7 self query per view cloud), zero
        elsewhere -- the caller sum-reduces.  With at least as many ranks as views (and a multiple
        of them) a view belongs to world/N ranks: each builds ONLY that view's grid and queries its
        share of that view's rows.  Otherwise every rank builds all N grids and queries rows
        [qlo, qhi) of each (padded) view cloud."""
        c, ss = self.comm, self.splat
        dev = pts_f.device
        N = len(lens)
        h = torch.zeros((tot,), dtype=torch.float32, device=dev)
        if N > 0 and c.world >= N and c.world % N == 0:
            per_view = c.world // N
            v, part = c.rank // per_view, c.rank % per_view
            lv = lens[v]
            qlo, qhi = shard_bounds(lv, per_view, part)
            if qhi > qlo:
                cloud = pts_f[fl[v]:fl[v] + lv].view(1, lv, 3)
                num_v = with_host_lengths(torch.tensor([lv], dtype=torch.int64, device=dev), [lv])
                r7 = torch.full((1,), float(ss.frnn_radius), dtype=torch.float32, device=dev)
                grid = frnn.build_grid(cloud, num_v, r7)
                qnum = with_host_lengths(torch.tensor([qhi - qlo], dtype=torch.int64, device=dev), [qhi - qlo])
                dists, _, _, _ = frnn.frnn_grid_points(cloud[:, qlo:qhi].contiguous(), cloud, qnum, num_v, K=7,
                                                       r=r7, grid=grid)
                qfirst = torch.tensor([fl[v] + qlo], dtype=torch.int64, device=dev)
                _lib.call("iso_splat_vrk_h", _lib.ptr(dists), _lib.ptr(qfirst), _lib.ptr(qnum), _lib.ptr(num_v),
                          _lib.ptr(h), 1, dists.shape[1], _lib.stream())
            return h
        mx = max(lens) if lens else 0
        padded = torch.zeros((N, max(mx, 1), 3), dtype=torch.float32, device=dev)
        for i in range(N):
            padded[i, :lens[i]] = pts_f[fl[i]:fl[i] + lens[i]]
        r7 = torch.full((N,), float(ss.frnn_radius), dtype=torch.float32, device=dev)
        grid = frnn.build_grid(padded, num, r7)
        qlo, qhi = shard_bounds(mx, c.world, c.rank)
        qlens = [min(max(l - qlo, 0), qhi - qlo) for l in lens]
        if qhi > qlo:
            qnum = with_host_lengths(torch.tensor(qlens, dtype=torch.int64, device=dev), qlens)
            dists, _, _, _ = frnn.frnn_grid_points(padded[:, qlo:qhi].contiguous(), padded, qnum, num, K=7,
                                                   r=r7, grid=grid)
            qfirst = torch.tensor([f + qlo for f in fl], dtype=torch.int64, device=dev)
            _lib.call("iso_splat_vrk_h", _lib.ptr(dists), _lib.ptr(qfirst), _lib.ptr(qnum), _lib.ptr(num),
                      _lib.ptr(h), N, dists.shape[1], _lib.stream())
        return h

    def _setup_with_h(self, pts_f, nrm_f, h, first, num):
        rs = self.rs
        dev = pts_f.device
        tot = pts_f.shape[0]
        N = self.views.shape[0]
        lens = num._iso_host
        ndc = torch.empty((tot, 3), dtype=torch.float32, device=dev)
        ellipse = torch.empty((tot, 3), dtype=torch.float32, device=dev)
        cutoff = torch.empty((tot,), dtype=torch.float32, device=dev)
        radii = torch.empty((tot, 2), dtype=torch.float32, device=dev)
        scaler = torch.empty((tot,), dtype=torch.float32, device=dev)
        p = _lib.ptr
        _lib.call("iso_splat_setup", p(pts_f), p(nrm_f), p(h), p(first), p(num), p(self.views), p(self.projs),
                  N, max(lens) if lens else 0, int(rs.image_size), float(rs.antialiasing_sigma),
                  float(rs.cutoff_threshold), p(ndc), p(ellipse), p(cutoff), p(radii), p(scaler), _lib.stream())
        return ndc, {"radii": radii, "ellipse_params": ellipse, "cutoff_threshold": cutoff, "scaler": scaler}

    def band_rows(self):
        """Output-image pixel rows [y0, y1) of this rank's tile-row band (the image is flipped)."""
        S = int(self.rs.image_size)
        b0, b1 = self.band
        return max(S - 16 * b1, 0), S - 16 * b0

    # -- stage 4: compositing + loss gradient + backward ------------------------------------------
    def composite_band(self, frags, filt):
        """(N,S,S,C+1) image, own band filled (renderer.py:53-78)."""
        idx, qv, occ = frags.idx, frags.qvalue, frags.occupancy
        N, S, _, K = idx.shape
        feat = filt["features"]
        C = feat.shape[1]
        y0, y1 = self.band_rows() if self.comm.world > 1 else (0, S)
        img = torch.zeros((N, S, S, C + 1), dtype=torch.float32, device=idx.device)
        p = _lib.ptr
        for n in range(N):
            if y1 > y0:
                _lib.call("iso_splat_composite", p(idx[n, y0:y1]), p(qv[n, y0:y1]), p(occ[n, y0:y1]),
                          p(filt["scaler"]), p(feat), (y1 - y0) * S, K, C, 1, 1e-4, None, p(img[n, y0:y1]),
                          _lib.stream())
        return img

    def backward(self, frags, filt, occ_grad_band, zbuf_grad_band):
        """occ_grad / zbuf_grad are valid on this rank's band (zero elsewhere).  Returns
        (grad (tot,3): xy rows of this rank's point slices + z of all points, visible flags)."""
        c = self.comm
        idx = frags.idx
        N, S, _, K = idx.shape
        dev = idx.device
        first, num = filt["first_idx"], filt["num_points"]
        lens, fl = num._iso_host, first._iso_host
        tot = filt["ndc"].shape[0]
        y0, y1 = self.band_rows() if c.world > 1 else (0, S)
        occ_grad = c.all_reduce_(occ_grad_band.contiguous(), "sum")
        if c.world == 1:
            vis, rs_ = _visible_and_radius(idx, filt["radii"], first, num, float(self.rs.radii_backward_scaler))
            grad = _C._backward(filt["ndc"], filt["radii"], occ_grad, first, num, visible=vis, rs=rs_, idx=idx,
                                grad_zbuf=zbuf_grad_band)
            return grad, vis
        # visible flags from the own band, max-reduced; median radius replicated
        vis = torch.zeros((tot,), dtype=torch.uint8, device=dev)
        p = _lib.ptr
        for n in range(N):
            if y1 > y0:
                _lib.call("iso_splat_mark_visible", p(idx[n, y0:y1]), (y1 - y0) * S, K, p(vis), _lib.stream())
        vis_i = vis.to(torch.int32)
        c.all_reduce_(vis_i, "max")
        vis = vis_i.to(torch.uint8)
        rs_ = median_radius(vis, filt["radii"], first, num, float(self.rs.radii_backward_scaler))
        # xy: point-major over this rank's slice of every view
        sub = [shard_bounds(l, c.world, c.rank) for l in lens]
        sfirst = [f + lo for f, (lo, hi) in zip(fl, sub)]
        snum = [hi - lo for lo, hi in sub]
        sf = with_host_lengths(torch.tensor(sfirst, dtype=torch.int64, device=dev), sfirst)
        sn = with_host_lengths(torch.tensor(snum, dtype=torch.int64, device=dev), snum)
        grad = _C._backward(filt["ndc"], filt["radii"], occ_grad, sf, sn, visible=vis, rs=rs_)
        # z: pixel-major scatter on the own band, sum-reduced
        gz = torch.zeros((tot, 1), dtype=torch.float32, device=dev)
        for n in range(N):
            if y1 > y0:
                _C._backward_zbuf(idx[n:n + 1, y0:y1], zbuf_grad_band[n:n + 1, y0:y1], gz)
        c.all_reduce_(gz, "sum")
        grad[:, 2] = gz[:, 0]
        return grad, vis

    # -- the whole cycle -----------------------------------------------------------------------
    def step(self):
        c = self.comm
        r1 = self.project_resample()
        pts_all = c.all_gather_rows(r1.points[0], self.P)
        nrm_all = c.all_gather_rows(r1.normals[0], self.P)
        feats = 0.5 * (torch.nn.functional.normalize(nrm_all, dim=-1) + 1)
        frags, filt = self.splat_forward(pts_all, nrm_all, feats)
        img = self.composite_band(frags, filt)
        # loss of SURVEY 8(d) cfg 3: mean((alpha - target)^2) [+ 1e-2 mean(rgb^2): no grad to the op]
        alpha = img[..., 3]
        N, S = alpha.shape[0], alpha.shape[1]
        y0, y1 = self.band_rows() if c.world > 1 else (0, S)
        occ_grad = torch.zeros_like(alpha)
        tgt = self.target if self.target is not None else torch.zeros_like(alpha)
        occ_grad[:, y0:y1] = 2.0 * (alpha[:, y0:y1] - tgt[:, y0:y1]) / alpha.numel()
        zbuf_grad = torch.zeros_like(frags.zbuf)
        zbuf_grad[:, y0:y1, :, 0] = 1e-3 / alpha.numel()
        grad, vis = self.backward(frags, filt, occ_grad, zbuf_grad)
        return r1, img, grad, frags, filt


def sphere_silhouette(S, n_views, dist, fov_deg, device):
    """Analytic alpha target of a unit sphere centred at the origin (SURVEY 8(d) cfg 3)."""
    ax = -1 + (2 * torch.arange(S, device=device) + 1.0) / S
    rr = (1.0 / math.sqrt(dist * dist - 1.0)) / math.tan(math.radians(fov_deg) / 2)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    return ((xx ** 2 + yy ** 2) <= rr ** 2).float()[None].expand(n_views, S, S).contiguous()
