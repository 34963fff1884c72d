"""Drop-in for `prefix_sum.prefix_sum_cuda` (built from FRNN/external/prefix_sum in the
reference's install recipe, README.md:36-41; call sites DSS/core/rasterizer.py:873,915).

prefix_sum_cuda(cnt, total, off): exclusive scan of the first `total` int32 entries
of `cnt` into `off` (both 1-D, caller allocated).
"""
import torch

from . import _lib


def prefix_sum_cuda(grid_cnt, num_grids, grid_off):
    n = int(num_grids)
    if n == 0:
        return grid_off
    if grid_cnt.dtype != torch.int32 or grid_off.dtype != torch.int32:
        raise TypeError("prefix_sum_cuda expects int32 tensors")
    lib = _lib.load()
    ws_bytes = lib.iso_prefix_sum_workspace_bytes(n, 1)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=grid_cnt.device)
    _lib.call("iso_prefix_sum", _lib.ptr(grid_cnt), _lib.ptr(grid_off), n, 1, n, _lib.ptr(ws),
              ws_bytes, _lib.stream())
    return grid_off
