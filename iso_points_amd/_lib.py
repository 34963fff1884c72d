"""ctypes binding of libisopoints_hip.so (C ABI: include/isopoints.h).

The library is the product; this module only moves pointers.  There is NO CPU
fallback: if the shared object is missing or an argument lives on the CPU the
call raises.  torch is used for device memory and the current HIP stream only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libisopoints_hip.so")
if os.environ.get("ISO_DEV_LIB"):        # development aid (tools/build_variant.sh): another build of the same library
    LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float

# name -> (restype, argtypes); must list every symbol include/isopoints.h declares
SIGNATURES = {
    "iso_version": (_c.c_char_p, []),
    "iso_last_error": (_c.c_char_p, []),
    "iso_project_sphere": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
    "iso_project_sphere_follow": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _P]),
    "iso_siren_raw_floats": (_L, [_I, _I]),
    "iso_siren_packed_floats": (_L, [_I, _I]),
    "iso_siren_set_gemm_mode": (_I, [_I]),
    "iso_siren_get_gemm_mode": (_I, []),
    "iso_siren_set_tail_from": (_I, [_I]),
    "iso_siren_set_drawn_tiles": (_I, [_I]),
    "iso_idr_set_drawn_tiles": (_I, [_I]),
    "iso_siren_step_launches": (_I, [_I, _I, _I]),
    "iso_siren_pack_weights": (_I, [_P, _P, _I, _I, _P]),
    "iso_project_siren_workspace_bytes": (_L, [_L, _I, _I]),
    "iso_project_siren_counts_offset": (_L, [_L, _I, _I]),
    "iso_project_siren": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _F, _F, _I, _F, _P, _L, _P]),
    "iso_siren_sdf_grad": (_I, [_P, _P, _P, _L, _P, _I, _I, _F, _F, _P, _L, _P]),
    "iso_idr_raw_floats": (_L, [_I, _I, _I, _I]),
    "iso_idr_packed_floats": (_L, [_I, _I]),
    "iso_idr_pack_weights": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "iso_project_idr_workspace_bytes": (_L, [_L, _I, _I]),
    "iso_project_idr": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _I, _F, _P, _L, _P]),
    "iso_idr_sdf_grad": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _P, _L, _P]),
    "iso_trace_sphere": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _I, _F, _P]),
    "iso_trace_siren": (_I, [_P, _P, _P, _P, _P, _L, _P, _I, _I, _F, _F, _F, _F, _I, _F, _P, _L, _P]),
    "iso_trace_idr": (_I, [_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _F, _F, _I, _F, _P, _L, _P]),
    "iso_ray_nearest_point_workspace_bytes": (_L, [_L]),
    "iso_ray_nearest_point": (_I, [_P, _L, _F, _F, _F, _P, _L, _P, _P, _P, _P, _L, _P]),
    "iso_raymarch_settle": (_I, [_P, _P, _L, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P]),
    "iso_raymarch_overshoot": (_I, [_P, _P, _L, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "iso_raymarch_secant": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "iso_image_sample": (_I, [_P, _I, _I, _I, _I, _P, _L, _I, _P, _P]),
    "iso_ear_candidates": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "iso_points_bbox": (_I, [_P, _P, _I, _L, _P, _P]),
    "iso_frnn_make_grid": (_I, [_P, _P, _P, _I, _L, _I, _P, _P]),
    "iso_frnn_make_grid_density": (_I, [_P, _P, _P, _I, _L, _I, _F, _P, _P]),
    "iso_frnn_insert_points": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _L, _I, _P]),
    "iso_prefix_sum_workspace_bytes": (_L, [_L, _I]),
    "iso_prefix_sum": (_I, [_P, _P, _L, _I, _L, _P, _L, _P]),
    "iso_frnn_scan_cells": (_I, [_P, _P, _P, _I, _L, _I, _P, _L, _P]),
    "iso_frnn_counting_sort": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _I, _P]),
    "iso_frnn_query_workspace_bytes": (_L, [_I, _L, _L]),
    "iso_frnn_query": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _L, _L, _L, _P, _L, _P]),
    "iso_frnn_gather": (_I, [_P, _P, _P, _I, _L, _L, _I, _I, _P]),
    "iso_repulse": (_I, [_P, _P, _P, _L, _P, _L, _L, _I, _P, _P]),
    "iso_upsample_candidates": (_I, [_P, _P, _L, _I, _P, _P, _P]),
    "iso_farthest_point_sampling_work_floats": (_L, [_I, _L]),
    "iso_farthest_point_sampling": (_I, [_P, _P, _P, _P, _I, _L, _L, _P, _P, _P]),
    "iso_splat_view_flags": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _I, _P]),
    "iso_compact_rows": (_I, [_P, _P, _P, _P, _L, _L, _I, _P]),
    "iso_splat_vrk_h": (_I, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "iso_splat_setup": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "iso_splat_tiles_per_side": (_I, [_I]),
    "iso_splat_bin_count": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P, _P]),
    "iso_splat_tile_offsets": (_I, [_P, _P, _P, _L, _P]),
    "iso_splat_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _F, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P,
                               _L, _P]),
    "iso_splat_forward_workspace_bytes": (_L, [_L, _I]),
    "iso_splat_render": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _F, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P,
                              _L, _P, _P, _I, _I, _F, _P, _P]),
    "iso_splat_render_visible": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _F, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P,
                                      _L, _P, _P, _I, _I, _F, _P, _P, _P]),
    "iso_gather_neg_idx": (_I, [_P, _P, _L, _P, _P]),
    "iso_splat_composite": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P, _P, _P]),
    "iso_splat_composite_backward": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "iso_splat_mark_visible": (_I, [_P, _L, _I, _P, _P]),
    "iso_splat_median_radius_workspace_bytes": (_L, [_I]),
    "iso_splat_median_radius": (_I, [_P, _P, _P, _P, _I, _L, _F, _P, _L, _P, _P]),
    "iso_splat_backward_workspace_bytes": (_L, [_I, _I, _I, _L]),
    "iso_splat_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _P, _P, _P, _I, _I, _I, _I, _F, _L, _P, _L, _P, _P]),
    "iso_splat_zbuf_backward": (_I, [_P, _P, _L, _I, _P, _P]),
    "iso_bricks_workspace_bytes": (_L, [_L]),
    "iso_bricks_workspace_init": (_I, [_P, _L, _P]),
    "iso_bricks_workspace_check": (_I, [_P, _L, _P]),
    "iso_bricks_build_whole": (_I, [_P, _P, _P, _L, _F, _I, _F, _P, _L, _P]),
    "iso_bricks_build_pending": (_I, [_P, _P, _P, _L, _F, _I, _F, _P, _L, _P, _P]),
    "iso_bricks_box_take": (_I, [_P, _L, _P, _P]),
    "iso_bricks_build": (_I, [_P, _P, _P, _L, _L, _P, _P, _P, _L, _P, _L, _F, _I, _F, _P, _L, _P]),
    "iso_bricks_params": (_I, [_P, _I, _L, _L, _L, _F, _I, _F, _P, _L, _P]),
    "iso_halo_export": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _P, _L, _P]),
    "iso_halo_import": (_I, [_P, _L, _P, _P, _I, _I, _I, _L, _P, _P, _P, _L, _P]),
    "iso_resample_fused": (_I, [_P, _L, _P, _L, _I, _P, _P, _P, _P]),
    "iso_splat_view_mask": (_I, [_P, _P, _P, _I, _L, _F, _F, _I, _P, _P, _P]),
    "iso_splat_h_fused": (_I, [_P, _L, _P, _P, _L, _P, _I, _P, _P]),
    "iso_splat_z_absmax": (_I, [_P, _L, _P, _P]),
    "iso_splat_z_scatter": (_I, [_P, _P, _L, _I, _L, _P, _P, _P]),
    "iso_splat_z_finish": (_I, [_P, _P, _L, _L, _P, _P]),
    "iso_splat_band_marks": (_I, [_P, _P, _I, _L, _L, _I, _P, _P, _P]),
    "iso_splat_band_z_scatter": (_I, [_P, _P, _I, _L, _L, _I, _P, _P, _P]),
    "iso_splat_repack": (_I, [_P, _L, _I, _I, _I, _P, _F, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "iso_splat_band_segment_floats": (_L, [_L]),
    "iso_splat_band_export_workspace_bytes": (_L, [_L, _I, _I]),
    "iso_splat_band_export": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _P, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P,
                                   _P, _L, _P]),
    "iso_splat_band_import": (_I, [_P, _I, _I, _L, _L, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "iso_splat_band_remap": (_I, [_P, _P, _I, _L, _L, _L, _I, _P, _P]),
    "iso_splat_band_return": (_I, [_P, _P, _P, _P, _P, _I, _L, _P, _P]),
    "iso_splat_band_merge": (_I, [_P, _P, _I, _I, _L, _L, _P, _P, _P, _P]),
    "iso_splat_median_pass_words": (_L, [_I]),
    "iso_splat_median_pass": (_I, [_I, _P, _P, _P, _P, _I, _L, _P, _L, _P]),
    "iso_splat_median_final": (_I, [_P, _I, _F, _P, _P]),
    "iso_insert_fathers": (_I, [_P, _P, _I, _L, _P, _P, _P, _P, _P]),
    "iso_insert_children": (_I, [_P, _P, _I, _L, _I, _I, _P, _P, _P, _P, _P]),
    "iso_splat_points_backward": (_I, [_P, _L, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "iso_rasterize_coarse": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "iso_rasterize_fine": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P]),
    "iso_splat_front_workspace_bytes": (_L, [_L]),
    "iso_splat_view_mask_scan": (_I, [_P, _P, _P, _I, _L, _F, _F, _I, _P, _P, _L, _P, _P, _P, _P]),
    "iso_splat_front_rows": (_I, [_P, _P, _P, _I, _I, _P, _P, _L, _P, _P, _I, _I, _F, _F, _P, _L, _P, _P, _P, _P, _P, _P,
                                  _P, _P, _P, _L, _P, _P]),
    "iso_splat_front": (_I, [_P, _P, _P, _I, _I, _P, _P, _L, _P, _P, _I, _I, _F, _F, _P, _L, _P, _P, _P, _P, _P, _P,
                             _P, _P, _P, _P, _P]),
}

class Follow(ctypes.Structure):
    """include/isopoints.h: iso_follow -- side work of a projection launch (host struct, read during the call)."""
    _fields_ = [("grid_ws", _P), ("grid_n_max", _L), ("views", _P), ("n_views", _I), ("znear", _F), ("zfar", _F),
                ("backface_culling", _I), ("mask_out", _P), ("front_ws", _P), ("front_ws_bytes", _L),
                ("first_idx_out", _P), ("num_pts_out", _P), ("view_total_out", _P)]


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "iso_points_amd: %s is missing -- build it with `make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no "
            "CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("iso_points_amd: tensor must live on the GPU (got %s); "
                           "there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("iso_points_amd: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        msg = load().iso_last_error().decode()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)
