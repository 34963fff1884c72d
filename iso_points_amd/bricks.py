"""Host side of the brick grid and its fused neighbour kernels (include/isopoints.h section E).

The cycle's two neighbour searches run here:
  resample_fused   UniformProjection.resample body, levelset_sampling.py:254-284
                   (= _create_tree :110-140 + the repulsion :268-284 in one kernel)
  splat_h_fused    K = 7 bandwidth of SurfaceSplatting._get_per_point_info, rasterizer.py:367-386,
                   for all views of the cloud in one pass
`frnn.frnn_grid_points` stays the general neighbour API.
"""
import torch

from . import _lib

import os as _os

# (ISO_RESAMPLE_CELL / ISO_H_CELL_SCALE: development overrides for parameter sweeps, tools/ab_cycle.sh)
RESAMPLE_CELL = float(_os.environ.get("ISO_RESAMPLE_CELL", "0.8"))   # fine cell = 0.8 r  (r = knn_k * sqrt(diag / P)): measured optimum at 1 M points
H_CELL_SCALE = float(_os.environ.get("ISO_H_CELL_SCALE", "6.0"))      # fine cell = 6 sqrt(diag / P) for the K = 7 bandwidth query: the cell then covers the 0.01 radius
                                                                     # inside which seven neighbours settle h (no uncertified counts: h + tail 272 -> 247 us at 1 M points)


def points_bbox(points, out=None):
    """(8,) f32 [min xyz, 0, max xyz, 0] of a packed (n,3) cloud, on the device (no host sync); out: where to write it."""
    pts = points.detach().float().contiguous().view(-1, 3)
    mm = out if out is not None else torch.empty((8,), dtype=torch.float32, device=pts.device)
    _lib.call("iso_points_bbox", _lib.ptr(pts), None, 1, pts.shape[0], _lib.ptr(mm), _lib.stream())
    return mm


class BrickGrid(object):
    """One workspace for clouds of up to n_own + import_max points on `device` (allocated once,
    rebuilt in place every cycle)."""

    def __init__(self, n_own, device, import_max=0):
        self.n_own, self.import_max = int(n_own), int(import_max)
        self.n_max = self.n_own + self.import_max
        lib = _lib.load()
        self.ws_bytes = lib.iso_bricks_workspace_bytes(self.n_max)
        self.ws = torch.empty((self.ws_bytes,), dtype=torch.uint8, device=device)
        assert self.ws.data_ptr() % 256 == 0
        _lib.call("iso_bricks_workspace_init", _lib.ptr(self.ws), self.n_max, _lib.stream())
        self.points = None
        self._seen = [0] * 16          # counter sums already reported by counters_since_last()

    def build(self, points, normals=None, payload=None, bbox=None, n_total=None, radius=-1.0, knn_k=8,
              cell_scale=None, id_base=0, imports=None, params_done=False, pending=False, follow=None):
        """points (n_own,3) f32 contiguous; imports = (rec0 (m,4), rec1 (m,4), count int32 (1,)) or None.
        params_done: the header was already written by iso_bricks_params (N ranks).
        pending (one rank, the whole cloud): the bounding box of `points` is the workspace's pending box, left there by
        the projection launch that wrote them (Follow); follow: that Follow when it also took the renderable mask --
        the scan of its counts is done on the side."""
        assert points.shape == (self.n_own, 3) and points.dtype == torch.float32 and points.is_contiguous()
        if cell_scale is None:
            cell_scale = RESAMPLE_CELL * knn_k
        p = _lib.ptr
        if pending:
            import ctypes
            assert imports is None and self.import_max == 0 and int(id_base) == 0
            fs = follow.struct() if (follow is not None and follow.views is not None) else None
            _lib.call("iso_bricks_build_pending", p(points), p(normals), p(payload), self.n_own, float(radius), int(knn_k),
                      float(cell_scale), p(self.ws), self.ws_bytes, ctypes.byref(fs) if fs is not None else None,
                      _lib.stream())
            self.points = points
            return self
        whole = (not params_done and bbox is None and imports is None and self.import_max == 0 and int(id_base) == 0
                 and (n_total is None or int(n_total) == self.n_own))
        if whole:                 # one rank, the whole cloud: the build takes the bounding box itself
            _lib.call("iso_bricks_build_whole", p(points), p(normals), p(payload), self.n_own, float(radius), int(knn_k),
                      float(cell_scale), p(self.ws), self.ws_bytes, _lib.stream())
            self.points = points
            return self
        if params_done:
            bbox, radius, knn_k, cell_scale = None, 1.0, 1, 1.0
        elif bbox is None:
            bbox = points_bbox(points)
        imp0 = imp1 = impc = None
        if imports is not None and self.import_max > 0:
            imp0, imp1, impc = imports
            assert imp0.shape[0] >= self.import_max and imp1.shape[0] >= self.import_max
        _lib.call("iso_bricks_build", p(points), p(normals), p(payload), self.n_own, int(id_base), p(imp0), p(imp1),
                  p(impc), self.import_max if imports is not None else 0, p(bbox),
                  int(n_total if n_total is not None else self.n_own), float(radius), int(knn_k), float(cell_scale),
                  p(self.ws), self.ws_bytes, _lib.stream())
        self.points = points
        self._imports = imports          # keep alive until the stream has used them
        return self

    # -- diagnostics (host sync) -------------------------------------------------------------------
    def counters_since_last(self):
        """The grid's 16 device-side counters summed over every grid built on this workspace since the previous call
        (current grid + the sticky sums the header writes keep, minus what was reported before)."""
        c = self.ws[256:384].cpu().view(torch.int32).tolist()
        # the device sums are int32 and wrap (slot 0 adds ~1e5 occupied bricks per grid): differences modulo 2^32
        tot = [(c[i] + c[16 + i]) & 0xffffffff for i in range(16)]
        out = [(tot[i] - self._seen[i]) & 0xffffffff for i in range(16)]
        self._seen = tot
        return out

    def check_initialised(self):
        """Raises when iso_bricks_workspace_init never ran on the workspace (host sync; include/isopoints.h)."""
        _lib.call("iso_bricks_workspace_check", _lib.ptr(self.ws), self.n_max, _lib.stream())

    def header(self):
        self.check_initialised()
        raw = self.ws[:128].cpu()
        f = raw.view(torch.float32).tolist()
        i = raw.view(torch.int32).tolist()
        c = self.ws[256:320].cpu().view(torch.int32).tolist()
        return {"f": f[8], "r": f[9], "inv_sigma": f[12], "diag": f[13], "nb": i[16:19], "n_bricks": i[19],
                "n": i[23], "n_own": i[24], "id_base": i[25], "g_covers_r": i[26], "occupied": c[0],
                "tail": c[1], "overflow_bricks": c[2], "tail_h": c[3]}


class Follow(object):
    """What a projection launch does on the side for the stages that consume its points (include/isopoints.h:
    iso_follow): the bounding box of the result goes to the PENDING BOX of `grid`'s workspace -- the grid is then built
    with grid.build(..., pending=True[, follow=this]) without a box pass -- and, with `views`, the renderable mask and
    the per-tile counts whose scan that build does on the side (`mask`, `total`, `scanned`: as view_mask_scan returns
    them, valid after the build).  Holds the tensors the calls write; `struct()` is the ctypes argument."""

    def __init__(self, grid, n, views=None, znear=1.0, zfar=100.0, backface_culling=True):
        dev = grid.ws.device
        self.grid, self.n = grid, int(n)
        self.done = False          # set by the projection: True when its route did the side work
        self.views = None
        self.mask = self.total = self.scanned = None
        self.znear, self.zfar, self.backface = float(znear), float(zfar), bool(backface_culling)
        if views is not None:
            nv = views.shape[0]
            self.views = views.detach().float().contiguous()
            self.mask = torch.empty((self.n,), dtype=torch.int32, device=dev)
            ws = torch.empty((_lib.load().iso_splat_front_workspace_bytes(self.n),), dtype=torch.uint8, device=dev)
            first = torch.empty((nv,), dtype=torch.int64, device=dev)
            num = torch.empty((nv,), dtype=torch.int64, device=dev)
            self.total = torch.empty((8,), dtype=torch.int32, device=dev)
            self.scanned = (ws, first, num, self.total)

    def struct(self):
        p = lambda t: (t.data_ptr() if t is not None else None)          # noqa: E731
        f = _lib.Follow()
        f.grid_ws, f.grid_n_max = p(self.grid.ws), self.grid.n_max
        f.views = p(self.views)
        f.n_views = self.views.shape[0] if self.views is not None else 0
        f.znear, f.zfar, f.backface_culling = self.znear, self.zfar, int(self.backface)
        if self.views is not None:
            ws, first, num, total = self.scanned
            f.mask_out, f.front_ws, f.front_ws_bytes = p(self.mask), p(ws), ws.numel()
            f.first_idx_out, f.num_pts_out, f.view_total_out = p(first), p(num), p(total)
        return f


def box_take(grid):
    """The pending box of the grid's workspace as (8,) f32 [min xyz, 0, max xyz, 0]; cleared."""
    mm = torch.empty((8,), dtype=torch.float32, device=grid.ws.device)
    _lib.call("iso_bricks_box_take", _lib.ptr(grid.ws), grid.n_max, _lib.ptr(mm), _lib.stream())
    return mm


def resample_fused(grid, k_plus_one, want_idx=False):
    """One repulsion move of the grid's own points -> (moved (n,3), idx (n,K) int64 or None, d2 or None)."""
    pts = grid.points
    n = grid.n_own
    out = torch.empty_like(pts)
    idx = d2 = None
    if want_idx:
        idx = torch.empty((n, k_plus_one - 1), dtype=torch.int64, device=pts.device)
        d2 = torch.empty((n, k_plus_one - 1), dtype=torch.float32, device=pts.device)
    p = _lib.ptr
    _lib.call("iso_resample_fused", p(grid.ws), grid.n_max, p(pts), n, int(k_plus_one), p(out), p(idx), p(d2),
              _lib.stream())
    return out, idx, d2


def view_mask(points, normals, views, znear=1.0, zfar=100.0, backface_culling=True):
    """-> mask (n,) int32 (bit v = renderable in view v), view_count (8,) int32 on the device."""
    n, nv = points.shape[0], views.shape[0]
    mask = torch.empty((n,), dtype=torch.int32, device=points.device)
    cnt = torch.empty((8,), dtype=torch.int32, device=points.device)
    p = _lib.ptr
    _lib.call("iso_splat_view_mask", p(points), p(normals), p(views), nv, n, float(znear), float(zfar),
              int(bool(backface_culling)), p(mask), p(cnt), _lib.stream())
    return mask, cnt


def view_mask_scan(points, normals, views, znear=1.0, zfar=100.0, backface_culling=True, total_out=None):
    """view_mask + the per-chunk counts and their scan the front end needs, in two launches (iso_splat_view_mask_scan):
    -> mask (n,) int32, view_total (8,) int32, and the `scanned` tuple (workspace, first_idx (N,) i64, num_points (N,)
    i64, view_total) SurfaceSplatting.front_setup takes to run its compaction + set-up pass alone."""
    n, nv = points.shape[0], views.shape[0]
    dev = points.device
    mask = torch.empty((n,), dtype=torch.int32, device=dev)
    ws = torch.empty((_lib.load().iso_splat_front_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    first = torch.empty((nv,), dtype=torch.int64, device=dev)
    num = torch.empty((nv,), dtype=torch.int64, device=dev)
    total = total_out if total_out is not None else torch.empty((8,), dtype=torch.int32, device=dev)   # (8,) int32, contiguous
    p = _lib.ptr
    _lib.call("iso_splat_view_mask_scan", p(points), p(normals), p(views), nv, n, float(znear), float(zfar),
              int(bool(backface_culling)), p(mask), p(ws), ws.numel(), p(first), p(num), p(total), _lib.stream())
    return mask, total, (ws, first, num, total)


def splat_h_fused(grid, mask, view_total, n_views):
    """h (n_views, n_own) f32: written where the mask bit is set, untouched elsewhere."""
    pts = grid.points
    h = torch.empty((n_views, grid.n_own), dtype=torch.float32, device=pts.device)
    p = _lib.ptr
    _lib.call("iso_splat_h_fused", p(grid.ws), grid.n_max, p(pts), p(mask), grid.n_own, p(view_total), int(n_views),
              p(h), _lib.stream())
    return h
