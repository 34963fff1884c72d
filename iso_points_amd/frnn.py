"""Drop-in for the `frnn` module the reference imports (lxxue/FRNN@eab337f).

Same call surface the reference uses:
  frnn.frnn_grid_points   DSS/models/levelset_sampling.py:132,182,200
                          DSS/utils/point_processing.py:74,84,145,179,253
                          DSS/core/rasterizer.py:371
  frnn.frnn_gather        DSS/models/levelset_sampling.py:213,268-271
  frnn._C.insert_points_cuda / counting_sort_cuda     DSS/core/rasterizer.py:909,921
All work is done by libisopoints_hip.so (include/isopoints.h section B).
"""
import ctypes

import os

import torch

from . import _lib

GRID_3D_PARAMS_SIZE = 8
GRID_2D_PARAMS_SIZE = 6
MAX_RES = 256                      # ISO_GRID_MAX_RES


def grid_max_res(p2):
    """Cells per axis the dense grid may use: 128 is plenty below ~250 k points, a 1 M-point
    surface wants 256 (4x fewer candidates per query); the arrays are (max_res+1)^3 ints."""
    if p2 <= 2048:
        return 16                     # 4 913 cells: small clouds (ray batches, tests) do not pay for a 2 M-cell table
    if p2 <= 32768:
        return 48
    if p2 <= 262144:
        return 128
    return 192 if p2 <= 655360 else MAX_RES


class FrnnGrid(object):
    """Opaque grid handle returned by frnn_grid_points (reusable for the same points2)."""

    def __init__(self, params, off, sorted_points, sorted_idx, lengths2, points2_shape, g_stride=None):
        self.g_stride = int(g_stride if g_stride is not None else off.shape[1])
        self.params = params
        self.off = off
        self.sorted_points = sorted_points
        self.sorted_idx = sorted_idx
        self.lengths2 = lengths2
        self.points2_shape = tuple(points2_shape)


def _as_radius(r, n, device):
    if torch.is_tensor(r):
        rt = r.detach().to(device=device, dtype=torch.float32).reshape(-1)
        if rt.numel() == 1 and n != 1:
            rt = rt.expand(n)
        return rt.contiguous()
    return torch.full((n,), float(r), dtype=torch.float32, device=device)


def _as_lengths(lengths, n, p, device):
    if lengths is None:
        return torch.full((n,), p, dtype=torch.int64, device=device)
    return lengths.to(device=device, dtype=torch.int64).contiguous()


def build_grid(points2, lengths2, radius, points_per_cell=8.0):
    """Grid build: make_grid -> insert -> scan -> counting sort (4 stages, no host sync).
    `points_per_cell`: density target of the cell size (see iso_frnn_make_grid_density)."""
    N, P2, D = points2.shape
    assert D == 3, "frnn_grid_points: only 3-D clouds are built here (2-D: use _C.*)"
    dev = points2.device
    params = torch.empty((N, GRID_3D_PARAMS_SIZE), dtype=torch.float32, device=dev)
    max_res = grid_max_res(P2)
    G = (max_res + 1) ** 3
    cnt = torch.zeros((N, G), dtype=torch.int32, device=dev)
    off = torch.empty((N, G), dtype=torch.int32, device=dev)
    cell = torch.empty((N, max(P2, 1)), dtype=torch.int32, device=dev)
    slot = torch.empty((N, max(P2, 1)), dtype=torch.int32, device=dev)
    sorted_pts = torch.empty((N, max(P2, 1), 3), dtype=torch.float32, device=dev)
    sorted_idx = torch.empty((N, max(P2, 1)), dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws_bytes = lib.iso_prefix_sum_workspace_bytes(G, N)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    s = _lib.stream()
    p = _lib.ptr
    if points_per_cell == 8.0 and os.environ.get("ISO_FRNN_PPC"):      # tuning hook (results do not depend on it)
        points_per_cell = float(os.environ["ISO_FRNN_PPC"])
    _lib.call("iso_frnn_make_grid_density", p(points2), p(lengths2), p(radius), N, P2, max_res,
              float(points_per_cell), p(params), s)
    _lib.call("iso_frnn_insert_points", p(points2), p(lengths2), p(params), p(cnt), p(cell), p(slot),
              N, P2, G, 3, s)
    _lib.call("iso_frnn_scan_cells", p(cnt), p(off), p(params), N, G, 3, p(ws), ws_bytes, s)
    _lib.call("iso_frnn_counting_sort", p(points2), p(lengths2), p(cell), p(slot), p(off),
              p(sorted_pts), p(sorted_idx), N, P2, G, 3, s)
    return FrnnGrid(params, off, sorted_pts, sorted_idx, lengths2, points2.shape, g_stride=G)


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None,
                     return_nn=False, return_sorted=True, radius_cell_ratio=2.0):
    """K nearest neighbours of points1 in points2 within radius r.

    Returns (dists (N,P1,K) squared / ascending / -1 padded, idxs (N,P1,K) int64 -1 padded,
    nn (N,P1,K,3) or None, grid)."""
    if points1.shape[0] != points2.shape[0]:
        raise ValueError("points1 and points2 must have the same batch dimension")
    if points1.shape[2] != points2.shape[2]:
        raise ValueError("dim of points1 and points2 do not match")
    if not points2.is_cuda:
        raise RuntimeError("iso_points_amd.frnn: tensors must be on the GPU; there is no CPU path")
    if K < 1 or K > 32:
        raise ValueError("K must be in [1, 32], got %d" % K)
    N, P1, _ = points1.shape
    P2 = points2.shape[1]
    dev = points2.device
    self_query = (points1 is points2) or (points1.data_ptr() == points2.data_ptr() and P1 == P2
                                          and lengths1 is lengths2)
    p1 = points1.detach().float().contiguous()
    p2 = p1 if self_query else points2.detach().float().contiguous()
    l2 = _as_lengths(lengths2, N, P2, dev)
    l1 = l2 if self_query else _as_lengths(lengths1, N, P1, dev)
    radius = _as_radius(r, N, dev)
    if grid is None:
        grid = build_grid(p2, l2, radius)
    dists = torch.empty((N, P1, K), dtype=torch.float32, device=dev)
    idxs = torch.empty((N, P1, K), dtype=torch.int64, device=dev)
    nn = torch.empty((N, P1, K, 3), dtype=torch.float32, device=dev) if return_nn else None
    if P1 > 0:
        if lengths1 is not None or (self_query and lengths2 is not None):
            # rows beyond lengths stay "not found"
            dists.fill_(-1.0)
            idxs.fill_(-1)
            if nn is not None:
                nn.zero_()
        p = _lib.ptr
        ws_bytes = _lib.load().iso_frnn_query_workspace_bytes(N, P1, P2)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _lib.call("iso_frnn_query", None if self_query else p(p1), None if self_query else p(l1),
                  p(p2), p(grid.sorted_points), p(grid.sorted_idx), p(l2), p(grid.off),
                  p(grid.params), p(radius), K, p(dists), p(idxs), p(nn) if nn is not None else None,
                  N, P1, P2, grid.g_stride, p(ws), ws_bytes, _lib.stream())
        grid.tail_counts = ws[: 4 * N].view(torch.int32)   # diagnostics: queries served by the tail kernel
    return dists, idxs, nn, grid


def frnn_gather(x, idxs, lengths=None):
    """x (N,P2,U), idxs (N,P1,K) -> (N,P1,K,U); rows with idx < 0 are zero."""
    N, P2, U = x.shape
    _, P1, K = idxs.shape
    xf = x.detach().float().contiguous()
    ii = idxs.to(torch.int64).contiguous()
    out = torch.empty((N, P1, K, U), dtype=torch.float32, device=x.device)
    _lib.call("iso_frnn_gather", _lib.ptr(xf), _lib.ptr(ii), _lib.ptr(out), N, P1, P2, K, U,
              _lib.stream())
    return out


class _CNamespace(object):
    """frnn._C: the two low-level entry points the splat backward re-uses for its
    2-D grid (DSS/core/rasterizer.py:906-929).  Outputs are caller-allocated."""

    @staticmethod
    def insert_points_cuda(points, lengths, grid_params, pc_grid_cnt, pc_grid_cell, pc_grid_idx, G):
        N, P, D = points.shape
        p = _lib.ptr
        _lib.call("iso_frnn_insert_points", p(points.contiguous()), p(lengths.to(torch.int64).contiguous()),
                  p(grid_params), p(pc_grid_cnt), p(pc_grid_cell), p(pc_grid_idx), N, P,
                  pc_grid_cnt.shape[1], D, _lib.stream())

    @staticmethod
    def counting_sort_cuda(points, lengths, pc_grid_cell, pc_grid_idx, pc_grid_off, points_sorted,
                           points_sorted_idxs):
        N, P, D = points.shape
        p = _lib.ptr
        _lib.call("iso_frnn_counting_sort", p(points.contiguous()), p(lengths.to(torch.int64).contiguous()),
                  p(pc_grid_cell), p(pc_grid_idx), p(pc_grid_off), p(points_sorted),
                  p(points_sorted_idxs), N, P, pc_grid_off.shape[1], D, _lib.stream())


_C = _CNamespace()
