"""Host-side mirror of the reference's splat rasteriser (DSS/core/rasterizer.py,
DSS/core/renderer.py and the pybind module DSS._C, DSS/csrc/ext.cpp:5-18).

  _C.splat_points / _splat_points_naive / _splat_points_occ_backward /
  _splat_points_occ_fast_cuda_backward / _backward_zbuf        ext.cpp:8-17
  rasterize_elliptical_points                                  rasterizer.py:678-740
  EllipticalRasterizer (autograd.Function)                     rasterizer.py:743-973
  PointsRasterizationSettings                                  rasterizer.py:39-100
  SurfaceSplatting.forward (filter -> per-point info -> NDC -> raster)  rasterizer.py:584-661
  SurfaceSplattingRenderer.forward (weights + compositing)     renderer.py:36-82

pytorch3d containers (Pointclouds, cameras) are out of scope: clouds are handed over as
`PackedClouds` (packed points + first_idx + num_points) and cameras as two (N,4,4) matrices
in pytorch3d's row-vector convention (world->view and full world->NDC).
All arithmetic runs in libisopoints_hip.so (include/isopoints.h section D).
"""
from typing import NamedTuple, Optional

import os
import torch
import torch.autograd as autograd

from . import _lib
from . import frnn
from .levelset_sampling import host_lengths, with_host_lengths

kMaxPointsPerPixel = 150     # as the reference (rasterization_utils.cuh:18); lists deeper than 32 take the slow kernel


def image_hw(image_size):
    """image_size as the reference takes it (an int: square) or pytorch3d's (H, W) pair -> (H, W).  Non-square
    images are beyond the reference's rasteriser (rasterizer.py:52); NDC then follows pytorch3d's convention:
    the shorter side spans [-1, 1], the longer [-e, e] with e = longer / shorter."""
    if isinstance(image_size, (tuple, list)):
        return int(image_size[0]), int(image_size[1])
    return int(image_size), int(image_size)


class PointFragments(NamedTuple):
    idx: torch.Tensor
    zbuf: torch.Tensor
    qvalue: torch.Tensor
    scaler: torch.Tensor
    occupancy: torch.Tensor


class PointsRasterizationSettings:
    """rasterizer.py:39-100 (same slots and defaults)."""
    __slots__ = ["cutoff_threshold", "backface_culling", "depth_merging_threshold", "Vrk_invariant",
                 "Vrk_isotropic", "radii_backward_scaler", "image_size", "points_per_pixel", "bin_size",
                 "max_points_per_bin", "clip_pts_grad", "antialiasing_sigma"]

    def __init__(self, backface_culling=True, cutoff_threshold=1, depth_merging_threshold=0.05,
                 Vrk_invariant=False, Vrk_isotropic=True, radii_backward_scaler=10, image_size=256,
                 points_per_pixel=8, bin_size=0, max_points_per_bin=None, clip_pts_grad=-1,
                 antialiasing_sigma=1.0):
        self.cutoff_threshold = cutoff_threshold
        self.backface_culling = backface_culling
        self.depth_merging_threshold = depth_merging_threshold
        self.Vrk_invariant = Vrk_invariant
        self.Vrk_isotropic = Vrk_isotropic
        self.radii_backward_scaler = radii_backward_scaler
        self.image_size = image_size
        self.points_per_pixel = points_per_pixel
        self.bin_size = bin_size
        self.max_points_per_bin = max_points_per_bin
        self.clip_pts_grad = clip_pts_grad
        self.antialiasing_sigma = antialiasing_sigma


class PackedClouds(object):
    """The three accessors rasterize_elliptical_points uses on its Pointclouds argument
    (rasterizer.py:700-702)."""

    def __init__(self, points_packed, first_idx, num_points):
        self._p, self._f, self._n = points_packed, first_idx, num_points

    def points_packed(self):
        return self._p

    def cloud_to_packed_first_idx(self):
        return self._f

    def num_points_per_cloud(self):
        return self._n


# ----------------------------------------------------------------------------- _C
def _f32c(t):
    return t.detach().float().contiguous()


def _i64c(t):
    return t.detach().to(torch.int64).contiguous()


def _max_pts(num_points):
    h = host_lengths(num_points)
    return max(h) if h else 0


class _CNamespace(object):
    """Drop-in for `from DSS import _C` (ext.cpp:5-18)."""

    @staticmethod
    def splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                     num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel,
                     bin_size=0, max_points_per_bin=0, tile_rows=None, out=None, max_pts=None,
                     pair_capacity=None, overflow_out=None, split_heavy_tiles=True, composite_with=None,
                     image_out=None, tile_cnt_ws=None, mark_visible=None):
        """-> (idx i32 (N,S,S,K), zbuf, qvalue f32 (N,S,S,K), occupancy f32 (N,S,S)).
        mark_visible (with composite_with): (P,) uint8, zero over the clouds' rows on entry -- the backward pass's
        visible flags, set while the lists are written (pass them on as _visible_and_radius(..., marked=True)).
        max_pts (upper bound of the points of a cloud) + pair_capacity (upper bound of the point-tile
        pairs): with both given nothing is read back to the host; an overflow of the pair list sets
        the int32 flag appended to `overflow_out` (checked by the caller when convenient).
        composite_with=(scaler (P,), features (P,C), norm_weighted, eps): also composite the image in the
        raster kernel (= composite(); iso_splat_render) into image_out (N,S,S,C+1): a 5th return value.
        bin_size / max_points_per_bin are accepted and ignored: binning is internal (16x16 tiles,
        exact-size pair list), so the reference's num_bins<22 / max_points_per_bin limits do not
        exist here.  tile_rows=(begin, end) (extension, used by the sharded path) rasterises
        only that band of 16-pixel tile rows (NDC pixel order) into `out` (or fresh -1/0 tensors)."""
        if not points.is_cuda:
            raise RuntimeError("iso_points_amd._C.splat_points: tensors must be on the GPU; there is no CPU path")
        K = int(points_per_pixel)
        S, W = image_hw(image_size)              # S = rows (H), W = columns
        if K > kMaxPointsPerPixel or K < 1:
            raise RuntimeError("Must have 1 <= points_per_pixel <= %d" % kMaxPointsPerPixel)
        if points.ndim != 2 or points.shape[1] != 3:
            raise RuntimeError("points must have shape (P, 3)")
        P = points.shape[0]
        if ellipse_params.shape != (P, 3) or radii.shape != (P, 2) or cutoff_thres.shape != (P,):
            raise RuntimeError("ellipse_params (P,3), radii (P,2), cutoff_thres (P,) expected")
        dev = points.device
        N = num_points_per_cloud.shape[0]
        done = getattr(num_points_per_cloud, "_iso_done", None)          # SurfaceSplatting.forward: rasterised already
        if done is not None:
            del num_points_per_cloud._iso_done
            if done[0] == (S, W, N, K, P) and tile_rows is None and out is None and composite_with is None:
                return done[1]
        pts, el, cu, ra = _f32c(points), _f32c(ellipse_params), _f32c(cutoff_thres), _f32c(radii)
        first, num = _i64c(cloud_to_packed_first_idx), _i64c(num_points_per_cloud)
        if hasattr(num_points_per_cloud, "_iso_host") and num is not num_points_per_cloud:
            with_host_lengths(num, host_lengths(num_points_per_cloud))
        lib = _lib.load()
        T = lib.iso_splat_tiles_per_side(S) if S > 0 else 0           # tile rows
        TW = lib.iso_splat_tiles_per_side(W) if W > 0 else 0          # tile columns
        band = (0, T) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
        if out is not None:
            idx, zbuf, qv, occ = out
        elif band == (0, T):
            idx = torch.empty((N, S, W, K), dtype=torch.int32, device=dev)
            zbuf = torch.empty((N, S, W, K), dtype=torch.float32, device=dev)
            qv = torch.empty((N, S, W, K), dtype=torch.float32, device=dev)
            occ = torch.empty((N, S, W), dtype=torch.float32, device=dev)
        else:
            idx = torch.full((N, S, W, K), -1, dtype=torch.int32, device=dev)
            zbuf = torch.full((N, S, W, K), -1.0, dtype=torch.float32, device=dev)
            qv = torch.full((N, S, W, K), -1.0, dtype=torch.float32, device=dev)
            occ = torch.zeros((N, S, W), dtype=torch.float32, device=dev)
        if N == 0 or S == 0 or W == 0:
            return idx, zbuf, qv, occ
        ntiles = N * T * TW
        maxp = int(max_pts) if max_pts is not None else _max_pts(num)
        p, s = _lib.ptr, _lib.stream()
        # counts -> offsets + cleared cursors in one launch; the counter array is cleared by the pass that reads it
        # (tile_cnt_ws: the caller's own zero-on-entry buffer of >= 4 (ntiles + 1) bytes -- owners that capture graphs keep
        #  theirs; default: one kept per device and stream)
        own_cnt = tile_cnt_ws is not None
        tile_cnt = (tile_cnt_ws[:4 * (ntiles + 1)] if own_cnt else _zeroed_workspace("tile_cnt", dev, 4 * (ntiles + 1))).view(torch.int32)
        tile_off = torch.empty((ntiles + 1,), dtype=torch.int32, device=dev)
        cursor = torch.empty((ntiles + 1,), dtype=torch.int32, device=dev)   # [ntiles] = overflow flag
        binned = getattr(num_points_per_cloud, "_iso_binned", None)      # SurfaceSplatting.forward: see prebin()
        if binned is not None and binned[0] == (S, W, band, N, P):
            tile_off, cursor, total = binned[1]
            del num_points_per_cloud._iso_binned
        else:
            try:
                _lib.call("iso_splat_bin_count", p(pts), p(ra), p(first), p(num), N, maxp, S, W, band[0], band[1],
                          p(tile_cnt), s)
                _lib.call("iso_splat_tile_offsets", p(tile_cnt), p(tile_off), p(cursor), ntiles + 1, s)
            except Exception:
                if not own_cnt:                                   # it may hold counts: never reuse it
                    for k in [k for k, v in _ZEROED.items() if v.data_ptr() == tile_cnt.data_ptr()]:
                        _ZEROED.pop(k, None)
                raise
            if pair_capacity is None:
                total = int(tile_off[ntiles].item())      # the one host read of the forward pass
            else:
                total = int(pair_capacity)
        pairs = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
        if overflow_out is not None:
            overflow_out.append(cursor[ntiles:])
        rws_b = lib.iso_splat_forward_workspace_bytes(N * TW * (band[1] - band[0]), K) if split_heavy_tiles else 0
        rws = torch.empty((rws_b,), dtype=torch.uint8, device=dev) if rws_b else None
        if composite_with is not None:
            sc_, ft_, norm_, eps_ = composite_with
            C = ft_.shape[1]
            img = image_out if image_out is not None else (
                torch.empty((N, S, W, C + 1), dtype=torch.float32, device=dev) if band == (0, T)
                else torch.zeros((N, S, W, C + 1), dtype=torch.float32, device=dev))
            # mark_visible: (P,) uint8, zero over the clouds' rows -- the lists' points are marked while they are written
            _lib.call("iso_splat_render_visible", p(pts), p(el), p(cu), p(ra), p(first), p(num), N, maxp,
                      float(depth_merging_thres), S, W, K, band[0], band[1], p(cursor), p(tile_off), p(pairs), total,
                      _lib.ctypes.c_void_p(cursor.data_ptr() + 4 * ntiles), p(idx), p(zbuf), p(qv), p(occ), p(rws), rws_b,
                      p(_f32c(sc_)), p(_f32c(ft_)), C, int(bool(norm_)), float(eps_), p(img), p(mark_visible), s)
            return idx, zbuf, qv, occ, img
        _lib.call("iso_splat_forward", p(pts), p(el), p(cu), p(ra), p(first), p(num), N, maxp,
                  float(depth_merging_thres), S, W, K, band[0], band[1], p(cursor), p(tile_off), p(pairs), total,
                  _lib.ctypes.c_void_p(cursor.data_ptr() + 4 * ntiles), p(idx), p(zbuf), p(qv), p(occ), p(rws), rws_b, s)
        return idx, zbuf, qv, occ

    @staticmethod
    def prebin(points, radii, first, num, max_pts, image_size):
        """The count pass and the tile offsets of splat_points on arrays whose row counts only the device knows yet
        (first / num (N,) int64 device tensors, max_pts >= every cloud's rows): SurfaceSplatting.forward issues them before
        its one host read, which then brings back the row counts AND the pair total (tile_off[-1]) -- splat_points no
        longer stops the queue between the offsets and the fill.  Returns (tile_off, cursor); attach
        ((S, W, band, N, rows), (tile_off, cursor, total)) to the num tensor splat_points will be given as `_iso_binned`."""
        S, W = image_hw(image_size)
        dev = points.device
        N = num.shape[0]
        lib = _lib.load()
        T, TW = lib.iso_splat_tiles_per_side(S), lib.iso_splat_tiles_per_side(W)
        ntiles = N * T * TW
        p, s = _lib.ptr, _lib.stream()
        tile_cnt = _zeroed_workspace("tile_cnt", dev, 4 * (ntiles + 1)).view(torch.int32)
        tile_off = torch.empty((ntiles + 1,), dtype=torch.int32, device=dev)
        cursor = torch.empty((ntiles + 1,), dtype=torch.int32, device=dev)
        try:
            _lib.call("iso_splat_bin_count", p(points), p(radii), p(first), p(num), N, int(max_pts), S, W, 0, T, p(tile_cnt), s)
            _lib.call("iso_splat_tile_offsets", p(tile_cnt), p(tile_off), p(cursor), ntiles + 1, s)
        except Exception:
            for k in [k for k, v in _ZEROED.items() if v.data_ptr() == tile_cnt.data_ptr()]:
                _ZEROED.pop(k, None)
            raise
        return tile_off, cursor, (S, W, (0, T), N)

    @staticmethod
    def _splat_points_naive(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                            num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel):
        return _CNamespace.splat_points(points, ellipse_params, cutoff_thres, radii,
                                        cloud_to_packed_first_idx, num_points_per_cloud,
                                        depth_merging_thres, image_size, points_per_pixel, 0, 0)

    @staticmethod
    def _backward(points, radii, grad_occ, first, num, visible=None, rs=None, rect_mode=0, radii_s=10.0,
                  idx=None, grad_zbuf=None, max_pts=None, rows_covered=False):
        """rows_covered: every row a caller will read lies in some cloud's [first, first + num) -- the kernels write
        all three components of every such row, so the result needs no zero fill (rows outside: unspecified)."""
        dev = points.device
        P = points.shape[0]
        N, S, W = grad_occ.shape[0], grad_occ.shape[1], grad_occ.shape[2]
        grad = (torch.empty if rows_covered else torch.zeros)((P, 3), dtype=torch.float32, device=dev)
        if P == 0 or N == 0:
            return grad
        pts, ra, go = _f32c(points), _f32c(radii), _f32c(grad_occ)
        first, num_c = _i64c(first), _i64c(num)
        if hasattr(num, "_iso_host") and num_c is not num:
            with_host_lengths(num_c, host_lengths(num))
        lib = _lib.load()
        ws_b = lib.iso_splat_backward_workspace_bytes(N, S, W, P)
        ws = torch.empty((max(ws_b, 1),), dtype=torch.uint8, device=dev)
        K = idx.shape[-1] if idx is not None else 1
        p = _lib.ptr
        _lib.call("iso_splat_backward", p(pts), p(ra), p(visible) if visible is not None else None,
                  p(_f32c(rs)) if rs is not None else None, p(first), p(num_c), N,
                  int(max_pts) if max_pts is not None else _max_pts(num_c), p(go),
                  p(idx.contiguous()) if idx is not None else None,
                  p(_f32c(grad_zbuf)) if grad_zbuf is not None else None, S, W, K, int(rect_mode),
                  float(radii_s), P, p(ws), ws.numel(), p(grad), _lib.stream())
        return grad

    @staticmethod
    def _splat_points_occ_backward(points, radii, grad_occ, cloud_to_packed_first_idx,
                                   num_points_per_cloud, radii_s, depth_merging_thres):
        """Slow rect-support occupancy backward (rasterize_points.cu:673-760) -> (P,2)."""
        g = _CNamespace._backward(points, radii, grad_occ, cloud_to_packed_first_idx, num_points_per_cloud,
                                  rect_mode=1, radii_s=radii_s)
        return g[:, :2].contiguous()

    @staticmethod
    def _splat_points_occ_fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ,
                                             num_points_per_cloud, cloud_to_packed_first_idx,
                                             points_grid_off=None, grid_params=None):
        """Disc-support occupancy backward (rasterize_points_backward.cu:30-212) -> (P,2).
        The 2-D grid arguments are accepted for signature compatibility and ignored: the kernel is
        point-major (each point walks its own pixel window), so no point grid is needed; the
        gradient of row i belongs to row i whatever order the caller sorted the points in."""
        g = _CNamespace._backward(points_sorted, radii_sorted, grad_occ, cloud_to_packed_first_idx,
                                  num_points_per_cloud, rs=rs)
        return g[:, :2].contiguous()

    @staticmethod
    def _backward_zbuf(idx, grad_zbuf, point_z_grad):
        """In-place: point_z_grad (P,1) += scatter of grad_zbuf (rasterize_points.cu:823-846)."""
        if idx.shape != grad_zbuf.shape or point_z_grad.ndim != 2 or point_z_grad.shape[1] != 1:
            raise RuntimeError("_backward_zbuf: idx/grad_zbuf (N,H,W,K) and point_z_grad (P,1) expected")
        if not point_z_grad.is_contiguous():
            raise RuntimeError("_backward_zbuf: point_z_grad must be contiguous")
        npix = idx.numel() // max(idx.shape[-1], 1)
        _lib.call("iso_splat_zbuf_backward", _lib.ptr(idx.contiguous()), _lib.ptr(_f32c(grad_zbuf)), npix,
                  idx.shape[-1], _lib.ptr(point_z_grad), _lib.stream())

    @staticmethod
    def _rasterize_coarse(points, radii, cloud_to_packed_first_idx, num_points_per_cloud, image_size, bin_size,
                          max_points_per_bin):
        """-> bin_points int32 (N, B, B, M), B = 1 + (image_size - 1) // bin_size (RasterizePointsCoarse,
        rasterize_points.h:167-214 / rasterize_points.cu:293-499): per bin the packed indices of the points with
        z >= 0 whose box overlaps the bin, ascending, -1 behind them.  Raises like the reference for >= 22 bins per
        side (rasterize_points.cu:462-468) and for a bin with more than max_points_per_bin points
        (rasterize_points_cpu.cpp:207-210; one host read of the flag)."""
        if not points.is_cuda:
            raise RuntimeError("iso_points_amd._C._rasterize_coarse: tensors must be on the GPU; there is no CPU path")
        S, bs, M = int(image_size), int(bin_size), int(max_points_per_bin)
        if bs <= 0 or S <= 0:
            raise RuntimeError("_rasterize_coarse: image_size and bin_size must be positive")
        B = 1 + (S - 1) // bs
        if B >= 22:
            raise RuntimeError("Got %d; that's too many!" % B)
        if points.ndim != 2 or points.shape[1] != 3 or radii.shape != (points.shape[0], 2):
            raise RuntimeError("points (P,3) and radii (P,2) expected")
        N = num_points_per_cloud.shape[0]
        dev = points.device
        bins = torch.empty((N, B, B, M), dtype=torch.int32, device=dev)
        ovf = torch.zeros((1,), dtype=torch.int32, device=dev)
        p = _lib.ptr
        _lib.call("iso_rasterize_coarse", p(_f32c(points)), p(_f32c(radii)), p(_i64c(cloud_to_packed_first_idx)),
                  p(_i64c(num_points_per_cloud)), N, S, bs, M, p(bins), p(ovf), _lib.stream())
        if M > 0 and int(ovf.item()):
            raise RuntimeError("Got too many points per bin")
        return bins

    @staticmethod
    def _rasterize_fine(points, ellipse_params, cutoff_thres, radii, bin_points, depth_merging_thres, image_size,
                        bin_size, points_per_pixel):
        """-> (idx, zbuf, qvalue, occupancy) from a bin table (RasterizePointsFine, rasterize_points.h:257-340 /
        rasterize_points.cu:503-673): _rasterize_fine(_rasterize_coarse(...)) == splat_points(...) bit for bit."""
        if not points.is_cuda:
            raise RuntimeError("iso_points_amd._C._rasterize_fine: tensors must be on the GPU; there is no CPU path")
        K, S, bs = int(points_per_pixel), int(image_size), int(bin_size)
        if K > kMaxPointsPerPixel or K < 1:
            raise RuntimeError("Must have num_closest <= %d" % kMaxPointsPerPixel)
        P = points.shape[0]
        if bin_points.ndim != 4 or bin_points.shape[1] != bin_points.shape[2] or bin_points.shape[1] != 1 + (S - 1) // bs:
            raise RuntimeError("bin_points (N, B, B, M) with B = 1 + (image_size - 1) // bin_size expected")
        if ellipse_params.shape != (P, 3) or radii.shape != (P, 2) or cutoff_thres.shape != (P,):
            raise RuntimeError("ellipse_params (P,3), radii (P,2), cutoff_thres (P,) expected")
        N, M = bin_points.shape[0], bin_points.shape[3]
        dev = points.device
        idx = torch.empty((N, S, S, K), dtype=torch.int32, device=dev)
        zbuf = torch.empty((N, S, S, K), dtype=torch.float32, device=dev)
        qv = torch.empty((N, S, S, K), dtype=torch.float32, device=dev)
        occ = torch.empty((N, S, S), dtype=torch.float32, device=dev)
        p = _lib.ptr
        _lib.call("iso_rasterize_fine", p(_f32c(points)), p(_f32c(ellipse_params)), p(_f32c(cutoff_thres)), p(_f32c(radii)),
                  P, p(bin_points.to(torch.int32).contiguous()), N, M, float(depth_merging_thres), S, bs, K, p(idx), p(zbuf),
                  p(qv), p(occ), _lib.stream())
        return idx, zbuf, qv, occ


_C = _CNamespace()


# ----------------------------------------------------------------------------- autograd op
def _visible_and_radius(idx, radii, first_idx, num_points, radii_s, max_pts=None, vis=None, med_ws=None, marked=False):
    """rasterizer.py:850-856,884: visible set + per-cloud r = median(visible radii) * radii_s,
    computed on the device (sort + device-side index, no host sync).  vis: (P,) uint8 whose rows of the clouds are
    already zero (the front end clears them: front_setup's "visible"); marked: the raster call has set the flags
    already (splat_points(mark_visible=vis)); med_ws: the caller's own zero-on-entry workspace for the median (see
    median_radius)."""
    P = radii.shape[0]
    dev = radii.device
    if vis is None:
        assert not marked
        vis = torch.zeros((P,), dtype=torch.uint8, device=dev)
    npix = idx.numel() // idx.shape[-1]
    if not marked:
        _lib.call("iso_splat_mark_visible", _lib.ptr(idx.contiguous()), npix, idx.shape[-1], _lib.ptr(vis),
                  _lib.stream())
    return vis, median_radius(vis, radii, first_idx, num_points, radii_s, max_pts=max_pts, ws=med_ws)


_ZEROED = {}


def _zeroed_workspace(tag, dev, nbytes):
    """A workspace for the entry points whose contract is "zero on entry, left zero on exit" (they clear what they
    have read instead of starting with a clearing pass): cleared once, when it is first made, and kept per device AND
    STREAM -- two calls on different streams would otherwise share counters in flight.  Never made while the stream is
    capturing (the buffer would live in the graph's private pool and be handed to eager code later): a caller that
    captures passes its own workspace (`ws=` of median_radius; IsoCycle does)."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (tag, dev.index, int(nbytes), int(stream))
    ws = _ZEROED.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("iso_points_amd: %s needs a zero-initialised workspace and none exists for this stream yet: "
                               "allocate one before capturing and pass it (ws=...)" % tag)
        ws = torch.zeros((int(nbytes),), dtype=torch.uint8, device=dev)
        _ZEROED[key] = ws
    return ws


def median_radius_workspace(n_clouds, dev):
    """A private zero-on-entry / zero-on-exit workspace for median_radius (owners that capture graphs keep their own)."""
    return torch.zeros((_lib.load().iso_splat_median_radius_workspace_bytes(int(n_clouds)),), dtype=torch.uint8, device=dev)


def median_radius(vis, radii, first_idx, num_points, radii_s, max_pts=None, ws=None):
    """(N,) device tensor r_n = median(visible radii of cloud n) * radii_s (iso_splat_median_radius).  ws: the caller's
    own workspace (median_radius_workspace); default: one kept per device and stream."""
    dev = radii.device
    N = num_points.shape[0]
    own = ws is not None
    if not own:
        ws = _zeroed_workspace("median_radius", dev, _lib.load().iso_splat_median_radius_workspace_bytes(N))
    ws_b = ws.numel()
    out = torch.empty((N,), dtype=torch.float32, device=dev)
    p = _lib.ptr
    try:
        _lib.call("iso_splat_median_radius", p(_f32c(radii)), p(vis), p(_i64c(first_idx)), p(_i64c(num_points)), N,
                  int(max_pts) if max_pts is not None else _max_pts(num_points), float(radii_s), p(ws), ws_b, p(out),
                  _lib.stream())
    except Exception:
        if not own:       # the histograms may be dirty: never reuse them
            for k in [k for k, v in _ZEROED.items() if v is ws]:
                _ZEROED.pop(k, None)
        raise
    return out


class EllipticalRasterizer(autograd.Function):
    """rasterizer.py:743-973.  backward returns d/d(pts_screen) only, from occ_grad (xy) and
    zbuf_grad (z); idx/qvalue gradients are ignored exactly as in the reference (:784)."""

    @staticmethod
    def forward(ctx, pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx,
                num_points_per_cloud, depth_merging_threshold, image_size, points_per_pixel,
                bin_size=0, max_points_per_bin=0, radii_backward_scaler=10.0):
        idx, zbuf, qvalue_map, occ_map = _C.splat_points(
            pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx,
            num_points_per_cloud, depth_merging_threshold, image_size, points_per_pixel, bin_size,
            max_points_per_bin)
        ctx.radii_backward_scaler = radii_backward_scaler
        ctx.depth_merging_threshold = depth_merging_threshold
        ctx.host = (host_lengths(cloud_to_packed_first_idx), host_lengths(num_points_per_cloud))
        if radii_backward_scaler == 0:
            raise NotImplementedError("radii_backward_scaler == 0 ('WeightBackward') does not exist in the "
                                      "reference either: its backward unpacks 8 saved tensors where 4 were "
                                      "saved (rasterizer.py:774-776 vs :807-809)")
        ctx.save_for_backward(pts_screen, radii, idx, cloud_to_packed_first_idx, num_points_per_cloud)
        # no gradient flows back through idx or qvalue: backward ignores qvalue_grad exactly as the reference does (:784) --
        # saying so here spares a compositor behind this op the gradient of its weights with respect to q (8 M terms)
        ctx.mark_non_differentiable(idx, qvalue_map)
        return idx, zbuf, qvalue_map, occ_map

    @staticmethod
    def backward(ctx, idx_grad, zbuf_grad, qvalue_grad, occ_grad):
        pts_screen, radii, idx, first_idx, num_points = ctx.saved_tensors
        with_host_lengths(first_idx, ctx.host[0])
        with_host_lengths(num_points, ctx.host[1])
        if occ_grad is None:
            occ_grad = torch.zeros(idx.shape[:3], dtype=torch.float32, device=idx.device)
        vis, rs = _visible_and_radius(idx, radii, first_idx, num_points, ctx.radii_backward_scaler)
        # the clouds tile the rows [0, P) (host lengths): every row is written by the kernels, no zero fill of the result
        fl, nl = ctx.host
        covered = sum(nl) == pts_screen.shape[0] and all(fl[i] == sum(nl[:i]) for i in range(len(nl)))
        grads = _C._backward(pts_screen, radii, occ_grad, first_idx, num_points, visible=vis, rs=rs,
                             idx=idx, grad_zbuf=zbuf_grad, rows_covered=covered)
        return (grads, None, None, None, None, None, None, None, None, None, None, None)


def rasterize_elliptical_points(pcls_screen, ellipse_params, cutoff_threshold, radii,
                                depth_merging_threshold: float = 0.05, image_size: int = 512,
                                points_per_pixel: int = 5, bin_size: Optional[int] = None,
                                max_points_per_bin: Optional[int] = None,
                                radii_backward_scaler: float = 10.0, clip_pts_grad: float = -1.0):
    """rasterizer.py:678-740."""
    points_packed = pcls_screen.points_packed()
    first = pcls_screen.cloud_to_packed_first_idx()
    num = pcls_screen.num_points_per_cloud()
    cutoff_threshold = cutoff_threshold.expand(points_packed.shape[0])
    if points_packed.requires_grad and clip_pts_grad > 0:
        def _clip(grad, value=clip_pts_grad):
            scaler = grad.norm(dim=-1, keepdim=True).clamp(0, value)
            return torch.nn.functional.normalize(grad, dim=-1) * scaler
        points_packed.register_hook(_clip)
    return EllipticalRasterizer.apply(points_packed, ellipse_params, cutoff_threshold, radii, first, num,
                                      depth_merging_threshold, image_size, points_per_pixel, bin_size or 0,
                                      max_points_per_bin or 0, radii_backward_scaler)


# ----------------------------------------------------------------------------- SurfaceSplatting
class SurfaceSplatting(object):
    """SurfaceSplatting.forward (rasterizer.py:584-661) on plain tensors.

    cameras = (views, projs): world->view and FULL world->NDC matrices, (N,4,4) each, row-vector
    convention; znear/zfar as on the reference's cameras (defaults 1, 100, :191-192)."""

    def __init__(self, cameras=None, raster_settings=None, frnn_radius=0.2, znear=1.0, zfar=100.0):
        self.cameras = cameras
        self.raster_settings = raster_settings or PointsRasterizationSettings()
        self.frnn_radius = frnn_radius
        self.znear, self.zfar = znear, zfar
        self._Vrk_h_value, self._Vrk_h_maker = None, None

    @property
    def _Vrk_h(self):
        """The K = 7 bandwidth of every packed row of the last forward / per_point_info call (rasterizer.py:367-386 keeps it
        on the object); after forward() it is gathered from the per-view bandwidths on first access."""
        if self._Vrk_h_value is None and self._Vrk_h_maker is not None:
            self._Vrk_h_value, self._Vrk_h_maker = self._Vrk_h_maker(), None
        return self._Vrk_h_value

    @_Vrk_h.setter
    def _Vrk_h(self, value):
        self._Vrk_h_value, self._Vrk_h_maker = value, None

    def filter_renderable(self, points, normals, views):
        """-> flags (N,P) int32, offsets (N*P) int32 exclusive scan, lens (host list)."""
        P, N = points.shape[0], views.shape[0]
        dev = points.device
        rs = self.raster_settings
        flags = torch.empty((N * P + 1,), dtype=torch.int32, device=dev)
        flags[N * P] = 0
        p = _lib.ptr
        _lib.call("iso_splat_view_flags", p(points), p(normals), p(views), p(flags), P, N,
                  float(self.znear), float(self.zfar), int(bool(rs.backface_culling)), _lib.stream())
        off = torch.empty_like(flags)
        lib = _lib.load()
        ws_b = lib.iso_prefix_sum_workspace_bytes(N * P + 1, 1)
        ws = torch.empty((ws_b,), dtype=torch.uint8, device=dev)
        _lib.call("iso_prefix_sum", p(flags), p(off), N * P + 1, 1, N * P + 1, p(ws), ws_b, _lib.stream())
        bounds = off[torch.arange(0, N + 1, device=dev) * P].tolist()      # host read (N+1 ints)
        lens = [bounds[i + 1] - bounds[i] for i in range(N)]
        return flags, off, lens

    def compact(self, x, flags, off, P, total_out):
        U = x.shape[1]
        out = torch.empty((total_out, U), dtype=torch.float32, device=x.device)
        _lib.call("iso_compact_rows", _lib.ptr(_f32c(x)), _lib.ptr(flags), _lib.ptr(off), _lib.ptr(out), P,
                  flags.numel() - 1, U, _lib.stream())
        return out

    def per_point_info(self, points_f, normals_f, first, num, views, projs):
        """_get_per_point_info + transform for already filtered packed clouds."""
        rs = self.raster_settings
        if rs.Vrk_invariant or not rs.Vrk_isotropic:
            raise NotImplementedError("only the default isotropic Vrk is built (SURVEY 2.1 #3)")
        dev = points_f.device
        lens = host_lengths(num)
        N, mx, tot = len(lens), (max(lens) if lens else 0), points_f.shape[0]
        p = _lib.ptr
        s = _lib.stream()
        # h: K=7 self query per filtered view cloud (rasterizer.py:367-386)
        padded = torch.zeros((N, max(mx, 1), 3), dtype=torch.float32, device=dev)
        firsts = host_lengths(first)
        for i in range(N):
            padded[i, :lens[i]] = points_f[firsts[i]:firsts[i] + lens[i]]
        dists, _, _, _ = frnn.frnn_grid_points(padded, padded, num, num, K=7, r=self.frnn_radius)
        h = torch.empty((tot,), dtype=torch.float32, device=dev)
        _lib.call("iso_splat_vrk_h", p(dists), p(first), p(num), None, p(h), N, dists.shape[1], s)
        self._Vrk_h = h
        ndc = torch.empty((tot, 3), dtype=torch.float32, device=dev)
        ellipse = torch.empty((tot, 3), dtype=torch.float32, device=dev)
        cutoff = torch.empty((tot,), dtype=torch.float32, device=dev)
        radii = torch.empty((tot, 2), dtype=torch.float32, device=dev)
        scaler = torch.empty((tot,), dtype=torch.float32, device=dev)
        _lib.call("iso_splat_setup", p(points_f), p(normals_f), p(h), p(first), p(num), p(_f32c(views)),
                  p(_f32c(projs)), N, mx, min(image_hw(rs.image_size)), float(rs.antialiasing_sigma),
                  float(rs.cutoff_threshold), p(ndc), p(ellipse), p(cutoff), p(radii), p(scaler), s)
        return ndc, {"radii": radii, "ellipse_params": ellipse, "cutoff_threshold": cutoff, "scaler": scaler}

    # -- fused front end (section E of the C ABI): no filtered copies, no host read ---------------------
    def front(self, points, normals, views, projs, features=None, features_from_normals=False, grid=None,
              out=None, capacity=None):
        """mask -> brick grid -> h (all views in one pass) -> compaction + per-point set-up of ONE cloud
        seen by N <= 8 cameras.  Returns a dict of capacity-sized packed arrays (rows beyond the
        device-side total are unspecified), first_idx / num_points int64 (N,) and view_total int32 (8,)
        on the device, mask (P,) int32, src (rows -> original point).  `out`: a (12, capacity) f32
        buffer to write ndc / ellipse / radii / scaler / features into (the multi-GPU wire layout)."""
        from . import bricks
        rs = self.raster_settings
        if rs.Vrk_invariant or not rs.Vrk_isotropic:
            raise NotImplementedError("only the default isotropic Vrk is built (SURVEY 2.1 #3)")
        P, N = points.shape[0], views.shape[0]
        if N > 8:
            raise NotImplementedError("front: at most 8 views per pass (SurfaceSplatting.forward runs it in chunks)")
        dev = points.device
        mask, cnt, scanned = bricks.view_mask_scan(points, normals, views, self.znear, self.zfar, rs.backface_culling)
        if grid is None:
            # kept between calls, one per cloud size (a workspace is laid out for ITS size and starts from cleared
            # counters: allocated and cleared once); a batch of clouds of different sizes alternates between a few
            grids = self.__dict__.setdefault("_grids", {})
            key = (P, str(dev))
            grid = grids.pop(key, None)
            if grid is None:
                while len(grids) >= 4:
                    grids.pop(next(iter(grids)))             # the least recently used size
                grid = bricks.BrickGrid(P, dev)
            grids[key] = grid                                # most recently used last
        grid.build(points, normals, payload=mask, radius=float(self.frnn_radius), cell_scale=bricks.H_CELL_SCALE)
        h = bricks.splat_h_fused(grid, mask, cnt, N)
        return self.front_setup(points, normals, views, projs, mask, h, features, features_from_normals, out, capacity,
                                scanned=scanned)

    def front_setup(self, points, normals, views, projs, mask, h, features=None, features_from_normals=False,
                    out=None, capacity=None, scanned=None):
        """Compaction + per-point set-up into packed rows.  scanned: the tuple bricks.view_mask_scan returned for this
        mask (the chunk counts are then already scanned: one launch instead of three)."""
        rs = self.raster_settings
        P, N = points.shape[0], views.shape[0]
        dev = points.device
        cap = int(capacity) if capacity is not None else max(N * P, 1)
        C = 3 if features_from_normals else (features.shape[1] if features is not None else 0)
        if out is None:
            out = torch.empty((12 * cap,), dtype=torch.float32, device=dev)
        flat = out.view(-1)
        assert flat.numel() >= 12 * cap
        ndc = flat[0:3 * cap].view(cap, 3)
        ellipse = flat[3 * cap:6 * cap].view(cap, 3)
        radii = flat[6 * cap:8 * cap].view(cap, 2)
        scaler = flat[8 * cap:9 * cap]
        if C == 3:
            feat = flat[9 * cap:12 * cap].view(cap, 3)
        else:
            feat = torch.empty((cap, C), dtype=torch.float32, device=dev) if C else None
        cutoff = torch.empty((cap,), dtype=torch.float32, device=dev)
        src = torch.empty((cap,), dtype=torch.int32, device=dev)
        p = _lib.ptr
        feat_in = p(_f32c(features)) if (features is not None and not features_from_normals) else None
        if scanned is not None:
            ws, first, num, view_total = scanned
            visible = torch.empty((cap,), dtype=torch.uint8, device=dev)       # rows of the clouds cleared by the launch
            ovf = getattr(self, "_row_overflow", None)                         # sticky flag: rows beyond `capacity` dropped
            if ovf is None or ovf.device != dev:
                ovf = self._row_overflow = torch.zeros((1,), dtype=torch.int32, device=dev)
            _lib.call("iso_splat_front_rows", p(points), p(normals), feat_in, C, int(bool(features_from_normals)), p(mask),
                      p(h), P, p(_f32c(views)), p(_f32c(projs)), N, min(image_hw(rs.image_size)),
                      float(rs.antialiasing_sigma), float(rs.cutoff_threshold), p(ws), ws.numel(), p(first), p(ndc),
                      p(ellipse), p(cutoff), p(radii), p(scaler), p(feat) if feat is not None else None, p(src),
                      p(visible), cap, p(ovf), _lib.stream())
            return {"ndc": ndc, "ellipse_params": ellipse, "cutoff_threshold": cutoff, "radii": radii, "scaler": scaler,
                    "features": feat, "src": src, "first_idx": first, "num_points": num, "view_total": view_total,
                    "mask": mask, "h": h, "wire": out, "capacity": cap, "visible": visible, "row_overflow": ovf}
        first = torch.empty((N,), dtype=torch.int64, device=dev)
        num = torch.empty((N,), dtype=torch.int64, device=dev)
        view_total = torch.empty((8,), dtype=torch.int32, device=dev)
        lib = _lib.load()
        ws_b = lib.iso_splat_front_workspace_bytes(P)
        ws = torch.empty((ws_b,), dtype=torch.uint8, device=dev)
        _lib.call("iso_splat_front", p(points), p(normals), p(_f32c(features)) if (features is not None and not features_from_normals) else None,
                  C, int(bool(features_from_normals)), p(mask), p(h), P, p(_f32c(views)), p(_f32c(projs)), N,
                  min(image_hw(rs.image_size)), float(rs.antialiasing_sigma), float(rs.cutoff_threshold), p(ws), ws_b, p(first),
                  p(num), p(view_total), p(ndc), p(ellipse), p(cutoff), p(radii), p(scaler),
                  p(feat) if feat is not None else None, p(src), _lib.stream())
        return {"ndc": ndc, "ellipse_params": ellipse, "cutoff_threshold": cutoff, "radii": radii, "scaler": scaler,
                "features": feat, "src": src, "first_idx": first, "num_points": num, "view_total": view_total,
                "mask": mask, "h": h, "wire": out, "capacity": cap}

    @staticmethod
    def _camera_matrices(cameras):
        """(views, projs) from a tuple of (N,4,4) tensors or from a pytorch3d-style camera object
        (get_world_to_view_transform() / get_full_projection_transform(), each with get_matrix(): the two
        matrices the reference reads, rasterizer.py:139,189,463-464)."""
        if isinstance(cameras, (tuple, list)):
            return cameras
        return (cameras.get_world_to_view_transform().get_matrix(), cameras.get_full_projection_transform().get_matrix())

    @staticmethod
    def _clouds_of(points, normals, features):
        """[(points (P,3), normals (P,3), features (P,C) or None)] of a tensor triple or a Pointclouds-like container
        (points_list() / normals_list() / features_list() when it has them, else the packed accessors with
        cloud_to_packed_first_idx() / num_points_per_cloud(); a container of one cloud needs only *_packed())."""
        if not hasattr(points, "points_packed"):
            return [(points, normals, features)]
        cloud = points
        B = len(cloud)
        if B == 1:
            f = features if features is not None else (cloud.features_packed() if hasattr(cloud, "features_packed") else None)
            return [(cloud.points_packed(), cloud.normals_packed(), f)]
        if hasattr(cloud, "points_list"):
            pl, nl = cloud.points_list(), cloud.normals_list()
            fl = cloud.features_list() if hasattr(cloud, "features_list") else None
            num = [int(x.shape[0]) for x in pl]
        else:
            first = [int(x) for x in cloud.cloud_to_packed_first_idx().tolist()]
            num = [int(x) for x in cloud.num_points_per_cloud().tolist()]
            pp, nn = cloud.points_packed(), cloud.normals_packed()
            ff = cloud.features_packed() if hasattr(cloud, "features_packed") else None
            pl, nl = [pp[f0:f0 + n] for f0, n in zip(first, num)], [nn[f0:f0 + n] for f0, n in zip(first, num)]
            fl = [ff[f0:f0 + n] for f0, n in zip(first, num)] if ff is not None else None
        if features is not None:
            # features= beside a container of B clouds: a list of B tensors, or ONE packed tensor in the container's
            # cloud order (split at the cloud lengths) -- anything else is an error, never silently dropped
            if isinstance(features, (list, tuple)):
                if len(features) != B or any(int(f.shape[0]) != n for f, n in zip(features, num)):
                    raise ValueError("features: expected %d tensors of %s rows" % (B, num))
                fl = list(features)
            else:
                if int(features.shape[0]) != sum(num):
                    raise ValueError("features: a packed tensor for %d clouds needs %d rows, got %d"
                                     % (B, sum(num), int(features.shape[0])))
                at = [sum(num[:b]) for b in range(B)]
                fl = [features[a:a + n] for a, n in zip(at, num)]
        return [(pl[b], nl[b], fl[b] if fl else None) for b in range(B)]

    def forward(self, points, normals=None, cameras=None, features=None):
        """SurfaceSplatting.forward (rasterizer.py:584-661).  points / normals: (P,3) tensors of ONE cloud, or a
        Pointclouds-like container of B clouds.  cameras: (views, projs) or a pytorch3d-style camera object, N of them.
        As in the reference one cloud is extended to the N cameras (:597-598, :229-241); B > 1 clouds need B cameras,
        cloud b is seen by camera b.  Any N (the fused front end takes 8 views per pass: it is run in chunks) and any
        feature width.  Returns (PointFragments, filtered dict): the packed rows are view-major / cloud-major in
        ascending point order -- the reference's packed layout."""
        clouds = self._clouds_of(points, normals, features)
        views, projs = self._camera_matrices(cameras if cameras is not None else self.cameras)
        views, projs = _f32c(views), _f32c(projs)
        rs = self.raster_settings
        N, B = views.shape[0], len(clouds)
        if B != 1 and B != N:
            raise ValueError("SurfaceSplatting.forward: %d clouds need %d cameras (or one cloud for any number), got %d"
                             % (B, B, N))
        dev = views.device
        (S, W), K = image_hw(rs.image_size), int(rs.points_per_pixel)
        # jobs: one cloud + a run of at most 8 cameras each (iso_splat_front's pass width)
        jobs = []
        if B == 1:
            for v0 in range(0, N, 8):
                jobs.append((clouds[0], v0, min(v0 + 8, N)))
        else:
            jobs = [(clouds[b], b, b + 1) for b in range(B)]
        parts = []
        for (pp, nn, ff), v0, v1 in jobs:
            pts, nrm = _f32c(pp.detach()), _f32c(nn.detach())
            wide = ff is not None and ff.shape[1] > 8        # wider than the front end packs: gathered afterwards
            with torch.no_grad():
                fr = self.front(pts, nrm, views[v0:v1], projs[v0:v1], features=None if wide else ff)
            parts.append((fr, pts, nrm, pp, ff if wide else None))
        # exact-size results: ONE host read of every job's row counts
        one = len(parts) == 1
        binned = None
        if one and min(S, W) > 0:
            # one job: the raster's count pass + tile offsets run on the front end's capacity-sized arrays BEFORE the host
            # read below, which then also brings the pair total (no second stop of the queue inside splat_points)
            fr0 = parts[0][0]
            with torch.no_grad():
                binned = _C.prebin(fr0["ndc"], fr0["radii"], fr0["first_idx"], fr0["num_points"], parts[0][1].shape[0],
                                   rs.image_size)
        early, ovf = None, []
        cap_pairs = int(getattr(self, "_pair_cap", 0))
        if binned is not None and cap_pairs > 0 and not os.environ.get("ISO_OPAPI_SYNC"):      # (the variable: A/B of the two orders)
            # ... and with a pair capacity known from earlier calls (1.25 x the largest total seen) the fill and the raster
            # themselves are ISSUED before the read, on the capacity-sized arrays (the kernels take the row counts from the
            # device): the host reads row counts, pair total and overflow flag while the GPU rasterises.  An overflow
            # (a frame with > 1.25 x the pairs of every earlier one) discards the result and takes the exact path.
            fr0 = parts[0][0]
            with torch.no_grad():
                fr0["num_points"]._iso_binned = (binned[2] + (int(fr0["ndc"].shape[0]),), (binned[0], binned[1], cap_pairs))
                early = _C.splat_points(fr0["ndc"], fr0["ellipse_params"], fr0["cutoff_threshold"], fr0["radii"],
                                        fr0["first_idx"], fr0["num_points"], rs.depth_merging_threshold, rs.image_size, K,
                                        max_pts=parts[0][1].shape[0], overflow_out=ovf)
        if one and binned is not None:
            extra = [binned[0][-1:].long()] + ([ovf[0].reshape(1).long()] if early is not None else [])
            both = torch.cat([parts[0][0]["num_points"]] + extra).tolist()          # ONE host read
            counts, total_pairs = both[:N], int(both[N])
            if early is not None and (int(both[N + 1]) != 0 or total_pairs > cap_pairs):
                early = None                                                        # pair list overflowed: exact path below
                with torch.no_grad():                                               # (the counters were consumed: count again)
                    fr0 = parts[0][0]
                    binned = _C.prebin(fr0["ndc"], fr0["radii"], fr0["first_idx"], fr0["num_points"], parts[0][1].shape[0],
                                       rs.image_size)
                    total_pairs = int(binned[0][-1].item())
            self._pair_cap = max(cap_pairs, int(1.25 * total_pairs) + 4096)
        else:
            counts = (parts[0][0]["num_points"] if one else torch.cat([fr["num_points"] for fr, _, _, _, _ in parts])).tolist()
        lens = [int(x) for x in counts]
        tot = sum(lens)
        if binned is not None:
            binned = (binned[2] + (tot,), (binned[0], binned[1], total_pairs))
        fl = [sum(lens[:i]) for i in range(N)]
        if one:                # the front end's own device-side layout (no host -> device copies of what the device has)
            num = with_host_lengths(parts[0][0]["num_points"], lens)
            first = with_host_lengths(parts[0][0]["first_idx"], fl)
            if early is not None and tot > 0:
                num._iso_done = ((S, W, N, K, tot), early)
            elif binned is not None and tot > 0:
                num._iso_binned = binned
        else:
            num = with_host_lengths(torch.tensor(lens, dtype=torch.int64, device=dev), lens)
            first = with_host_lengths(torch.tensor(fl, dtype=torch.int64, device=dev), fl)

        def make_flags():
            flags_jobs = [((fr["mask"][None] >> torch.arange(v1 - v0, device=dev)[:, None]) & 1).to(torch.int32)
                          for (fr, _, _, _, _), (_, v0, v1) in zip(parts, jobs)]
            if B == 1:
                return torch.cat(flags_jobs, dim=0) if len(flags_jobs) > 1 else flags_jobs[0]      # (N, P)
            pmax = max(int(f.shape[1]) for f in flags_jobs)          # (B, max P): a cloud's row is zero past its length
            flags = torch.zeros((B, pmax), dtype=torch.int32, device=dev)
            for b, f in enumerate(flags_jobs):
                flags[b, :f.shape[1]] = f[0]
            return flags

        if tot == 0:
            flags = make_flags()
            idx = torch.full((N, S, W, K), -1, dtype=torch.int32, device=dev)
            neg = torch.full((N, S, W, K), -1.0, dtype=torch.float32, device=dev)
            occ = torch.zeros((N, S, W), dtype=torch.float32, device=dev)
            return PointFragments(idx, neg, neg.clone(), neg.clone(), occ), {"num_points": num, "first_idx": first,
                                                                             "flags": flags}
        # Columns of the packed rows.  What the raster needs is built now; the rest of the reference's `filtered` dictionary
        # (gathered points / normals / wide features, int64 src, flags, visibility, the rows' bandwidths) on first access.
        cols = {k: [] for k in ("radii", "ellipse_params", "cutoff_threshold", "scaler", "ndc", "features")}
        spans = []                                                     # per job: (first view's offset into lens, views, rows)
        v_at = 0
        for (fr, pts, nrm, pp, wide_ff), (_, v0, v1) in zip(parts, jobs):
            nv = v1 - v0
            jl = lens[v_at:v_at + nv]
            jt = sum(jl)
            spans.append((v_at, nv, jt))
            for k in ("radii", "ellipse_params", "cutoff_threshold", "scaler"):
                cols[k].append(fr[k][:jt])
            ndc = fr["ndc"][:jt]
            if pp.requires_grad:
                # the reference's gradient reaches the world points through cameras.transform_points (:618); the
                # set-up is under no_grad (:608-610).  (first / num of the job's rows: the front end's device tensors)
                ndc = _WorldToRows.apply(pp, ndc, views[v0:v1], projs[v0:v1], fr["mask"], fr["src"],
                                         fr["first_idx"], fr["num_points"])
            cols["ndc"].append(ndc)
            if wide_ff is None and fr["features"] is not None:
                cols["features"].append(fr["features"][:jt])
            v_at += nv

        def src_of(j):
            return parts[j][0]["src"][:spans[j][2]].long()

        def gathered(which):
            def make():
                out = []
                for j, (fr, pts, nrm, pp, wide_ff) in enumerate(parts):
                    x = {"points": pts, "normals": nrm, "features": _f32c(wide_ff) if wide_ff is not None else None}[which]
                    out.append(x[src_of(j)])
                return torch.cat(out, dim=0) if len(out) > 1 else out[0]
            return make

        def make_h():
            # the bandwidth of every packed row: view v's rows are a slice (lengths known on the host), gathered view by
            # view (torch.repeat_interleave over 2 M rows was 0.2 ms of the operator-API cycle)
            out = []
            for j, (fr, _, _, _, _) in enumerate(parts):
                v_at, nv, jt = spans[j]
                jl = lens[v_at:v_at + nv]
                jf0 = [sum(jl[:i]) for i in range(nv)]
                src = src_of(j)
                out.append(torch.cat([fr["h"][v][src[jf0[v]:jf0[v] + jl[v]]] for v in range(nv)]) if nv > 1 else fr["h"][0][src])
            return torch.cat(out, dim=0) if len(out) > 1 else out[0]

        cat = {k: (torch.cat(v, dim=0) if len(v) > 1 else v[0]) if v else None for k, v in cols.items()}
        info = {k: cat[k] for k in ("radii", "ellipse_params", "cutoff_threshold", "scaler")}
        ndc = cat["ndc"]
        self._Vrk_h_maker, self._Vrk_h_value = make_h, None
        idx, zbuf, qv, occ = rasterize_elliptical_points(
            PackedClouds(ndc, first, num), info["ellipse_params"], info["cutoff_threshold"], info["radii"],
            depth_merging_threshold=rs.depth_merging_threshold, image_size=rs.image_size, points_per_pixel=K,
            bin_size=rs.bin_size, max_points_per_bin=rs.max_points_per_bin,
            radii_backward_scaler=rs.radii_backward_scaler, clip_pts_grad=rs.clip_pts_grad)
        frag_scaler = gather_with_neg_idx(info["scaler"], idx)
        frags = PointFragments(idx, zbuf, qv, frag_scaler, occ)

        def make_vis():
            vis = torch.zeros((tot,), dtype=torch.uint8, device=dev)
            _lib.call("iso_splat_mark_visible", _lib.ptr(idx), N * S * W, K, _lib.ptr(vis), _lib.stream())
            return vis.bool()

        makers = {"points": gathered("points"), "normals": gathered("normals"), "flags": make_flags, "visibility": make_vis,
                  "src": lambda: (torch.cat([src_of(j) for j in range(len(parts))]) if len(parts) > 1 else src_of(0))}
        eager = {"ndc": ndc, "num_points": num, "first_idx": first, **info}
        if any(wide is not None for _, _, _, _, wide in parts):
            makers["features"] = gathered("features")
        else:
            eager["features"] = cat["features"]
        return frags, LazyDict(eager, makers)


class _WorldToRows(autograd.Function):
    """The packed NDC rows as a function of the world points: forward hands the rows of the front end through,
    backward is iso_splat_points_backward (SurfaceSplatting.transform, rasterizer.py:565-582)."""

    @staticmethod
    def forward(ctx, points, ndc_rows, views, projs, mask, src, first, num):
        ctx.save_for_backward(points.detach(), views, projs, mask, src, first, num)
        return ndc_rows.clone()

    @staticmethod
    def backward(ctx, grad_rows):
        points, views, projs, mask, src, first, num = ctx.saved_tensors
        pts = _f32c(points)
        g = _f32c(grad_rows)
        out = torch.empty_like(pts)
        p = _lib.ptr
        _lib.call("iso_splat_points_backward", p(pts), pts.shape[0], p(views), p(projs), views.shape[0], p(mask), p(src),
                  p(first), p(num), p(g), p(out), _lib.stream())
        return out.to(points.dtype), None, None, None, None, None, None, None


def gather_with_neg_idx(values, idx):
    """utils/__init__.py:172-185: values[idx], 0 where idx < 0.  One float per point and int32 indices on the GPU (the
    fragments' scaler, rasterizer.py:635-637): one kernel (iso_gather_neg_idx) instead of five elementwise passes."""
    if (values.is_cuda and values.ndim == 1 and values.dtype == torch.float32 and idx.dtype == torch.int32
            and idx.is_contiguous() and not values.requires_grad):
        out = torch.empty(idx.shape, dtype=torch.float32, device=idx.device)
        _lib.call("iso_gather_neg_idx", _lib.ptr(values.contiguous()), _lib.ptr(idx), idx.numel(), _lib.ptr(out), _lib.stream())
        return out
    g = values[idx.long().clamp(min=0)]
    return torch.where(idx >= 0, g, torch.zeros_like(g))


class LazyDict(dict):
    """A dict whose listed entries are computed on first access (`makers`: key -> thunk).  SurfaceSplatting.forward
    returns the reference's `filtered` dictionary this way: a caller that only reads the scaler and the normals does
    not pay for the gathered points, the flags, the visibility marks ... (0.4 ms of torch glue at 2 M rows).  Iterating,
    copying or unpacking the dict materialises everything."""

    def __init__(self, eager, makers):
        super().__init__(eager)
        self._makers = dict(makers)

    def __missing__(self, key):
        mk = self._makers.pop(key, None)
        if mk is None:
            raise KeyError(key)
        v = mk()
        super().__setitem__(key, v)
        return v

    def _all(self):
        for k in list(self._makers):
            self[k]
        return self

    def __contains__(self, key):
        return super().__contains__(key) or key in self._makers

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return super(LazyDict, self._all()).keys()

    def values(self):
        return super(LazyDict, self._all()).values()

    def items(self):
        return super(LazyDict, self._all()).items()

    def __iter__(self):
        return super(LazyDict, self._all()).__iter__()

    def __len__(self):
        return super().__len__() + len(self._makers)

    def copy(self):
        return dict(self._all())

    # the rest of the plain-dict contract: these would by-pass __missing__ -- materialise first
    def pop(self, key, *default):
        if key in self._makers:
            self[key]
        return super().pop(key, *default)

    def popitem(self):
        return super(LazyDict, self._all()).popitem()

    def setdefault(self, key, default=None):
        if key in self._makers:
            return self[key]
        return super().setdefault(key, default)

    def __delitem__(self, key):
        if key in self._makers:
            del self._makers[key]
            if not super().__contains__(key):
                return
        super().__delitem__(key)

    def __eq__(self, other):
        return dict.__eq__(self._all(), other._all() if isinstance(other, LazyDict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class _Composite(autograd.Function):
    """iso_splat_composite with its backward: gradients to the per-point features and scaler, the
    fragments' qvalue and occupancy (idx carries none)."""

    @staticmethod
    def forward(ctx, idx, qvalue, occupancy, scaler, features, norm_weighted, eps):
        N, S, S2, K = idx.shape
        C = features.shape[1]
        out = torch.empty((N, S, S2, C + 1), dtype=torch.float32, device=idx.device)
        idx_c, qv, occ, sc, ft = idx.contiguous(), _f32c(qvalue), _f32c(occupancy), _f32c(scaler), _f32c(features)
        _lib.call("iso_splat_composite", _lib.ptr(idx_c), _lib.ptr(qv), _lib.ptr(occ), _lib.ptr(sc), _lib.ptr(ft),
                  N * S * S2, K, C, int(bool(norm_weighted)), float(eps), None, _lib.ptr(out), _lib.stream())
        ctx.save_for_backward(idx_c, qv, sc, ft)
        ctx.cfg = (int(bool(norm_weighted)), float(eps))
        return out

    @staticmethod
    def backward(ctx, grad_img):
        idx, qv, sc, ft = ctx.saved_tensors
        N, S, S2, K = idx.shape
        C = ft.shape[1]
        need = ctx.needs_input_grad
        if not (need[1] or need[3] or need[4]):
            # only the occupancy channel carries a gradient (features, scaler and q are constants of this graph -- the
            # splat op declares its qvalue non-differentiable): dL/docc is the image gradient's last channel, no kernel
            return None, None, (grad_img[..., C].contiguous() if need[2] else None), None, None, None, None
        g = _f32c(grad_img)
        gq = torch.empty_like(qv) if need[1] else None
        gocc = torch.empty((N, S, S2), dtype=torch.float32, device=idx.device) if need[2] else None
        gsc = torch.zeros_like(sc) if need[3] else None
        gft = torch.zeros_like(ft) if need[4] else None
        p = _lib.ptr
        _lib.call("iso_splat_composite_backward", p(idx), p(qv), p(sc), p(ft), p(g), N * S * S2, K, C, ctx.cfg[0],
                  ctx.cfg[1], p(gft), p(gq), p(gsc), p(gocc), _lib.stream())
        return None, gq, gocc, gsc, gft, None, None


def composite(fragments, scaler, features, norm_weighted=True, eps=1e-4):
    """SurfaceSplattingRenderer.forward (renderer.py:53-78): per-pixel weights exp(-q/2)*scaler,
    (norm-)weighted sum of per-point features, occupancy appended as alpha -> (N,S,S,C+1).
    `scaler` is the per-point EWA normaliser (P,), features (P,C) packed.  Differentiable with respect
    to features, scaler, fragments.qvalue and fragments.occupancy like the reference's compositor."""
    return _Composite.apply(fragments.idx, fragments.qvalue, fragments.occupancy, scaler, features, norm_weighted, eps)
