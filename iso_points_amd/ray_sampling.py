"""Ray-side sampling around the iso-points: the hot statements of
CombinedModel.sample_offsurface_using_isopoints (DSS/models/combined_modeling.py:317-386) on
plain tensors -- visible / occluded iso-points by splatting, the ray -> nearest-point search
that bounds the in-surface segment of every ray, and the lowest-SDF candidate on that segment.
Cameras, masks and the off-surface cube sampling around them stay with the caller (out of scope).
"""
import torch
import torch.nn.functional as F

from . import _lib
from .sdf_models import FusedSdf


def eps_sqrt(squared, eps=1e-17):
    """utils/mathHelper.py:21-25: clamp(|x|, eps) (the caller takes the root)."""
    return squared.abs().clamp_min(eps)


def ray_nearest_point(ray0, cam_pos, points):
    """For every ray from `cam_pos` (3,) with unit direction ray0 (R,3): the point of `points` (M,3)
    closest to the ray's line (combined_modeling.py:340-345 / :347-352).
    -> ray_sq (R,) = (pC . ray)^2 at that point, idx (R,) int64 (-1 if M == 0), dist (R,)."""
    if not ray0.is_cuda:
        raise RuntimeError("iso_points_amd: rays must be on the GPU; there is no CPU path")
    rays = ray0.detach().reshape(-1, 3).float().contiguous()
    pts = points.detach().reshape(-1, 3).float().contiguous()
    R, M, dev = rays.shape[0], pts.shape[0], rays.device
    c = [float(v) for v in cam_pos.reshape(3).tolist()]
    idx = torch.empty((R,), dtype=torch.int32, device=dev)
    ray_sq = torch.empty((R,), dtype=torch.float32, device=dev)
    dist = torch.empty((R,), dtype=torch.float32, device=dev)
    nb = _lib.load().iso_ray_nearest_point_workspace_bytes(R)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    p = _lib.ptr
    _lib.call("iso_ray_nearest_point", p(rays), R, c[0], c[1], c[2], p(pts) if M else None, M, p(idx), p(ray_sq),
              p(dist), p(ws), nb, _lib.stream())
    return ray_sq, idx.long(), dist


def insurface_segments(cam_pos, ray0, frontal_points, occluded_points):
    """combined_modeling.py:336-357 for one batch element: the near bound of a ray's in-surface
    segment is the projection of the closest frontal (visible) iso-point, the far bound that of the
    closest occluded one.  -> ray_len0 (R,), ray_len1 (R,) (already eps_sqrt().sqrt()), valid (R,) bool."""
    sq1, _, _ = ray_nearest_point(ray0, cam_pos, occluded_points)
    sq0, _, _ = ray_nearest_point(ray0, cam_pos, frontal_points)
    valid = sq0 < sq1
    return eps_sqrt(sq0).sqrt(), eps_sqrt(sq1).sqrt(), valid


def lowest_sdf_on_segments(model, cam_pos, cam_ray, ray_len0, ray_len1, n_points_per_ray=64):
    """combined_modeling.py:366-386: n_points_per_ray uniform candidates strictly inside
    [ray_len0, ray_len1] along each ray, the value-only fused SDF on all of them, the candidate with
    the lowest value per ray.  cam_pos (P,3) or (3,), cam_ray (P,3) unit -> p_insurface (P,3)."""
    sdf = model if isinstance(model, FusedSdf) else FusedSdf(model, cam_ray.device)
    lin = torch.linspace(0, 1.0, n_points_per_ray + 2, device=cam_ray.device)[1:-1]
    lengths = lin * (ray_len1 - ray_len0).view(-1, 1) + ray_len0.view(-1, 1)                      # (P, n)
    cand = lengths.unsqueeze(-1) * cam_ray.unsqueeze(-2) + cam_pos.reshape(-1, 1, 3)              # (P, n, 3)
    val = sdf(cand.reshape(-1, 3)).view(-1, n_points_per_ray)
    p_idx = torch.argmin(val, dim=-1, keepdim=True)
    return torch.gather(cand, -2, p_idx.unsqueeze(-1).expand(-1, -1, 3)).squeeze(-2)


def get_visible_points(points, normals, cameras, depth_merge_threshold=0.05, return_mask=False, znear=1.0,
                       zfar=100.0):
    """utils/__init__.py:699-711: splat one cloud (P,3) at 256 x 256 for every camera
    (cameras = (views, projs), (N,4,4) each, see rasterizer.SurfaceSplatting) and keep the points
    that own a fragment.  -> list of (n_i,3) visible points per camera [, mask (N,P) bool]."""
    from .rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from .levelset_sampling import host_lengths
    rs = PointsRasterizationSettings(depth_merging_threshold=depth_merge_threshold, image_size=256,
                                     cutoff_threshold=1.0, backface_culling=True)
    sp = SurfaceSplatting(cameras, rs, znear=znear, zfar=zfar)
    _, f = sp.forward(points, normals)
    N, P = cameras[0].shape[0], points.shape[0]
    lens, firsts = host_lengths(f["num_points"]), host_lengths(f["first_idx"])
    if "visibility" not in f:
        vis_list = [points.new_zeros((0, 3)) for _ in range(N)]
        mask = torch.zeros((N, P), dtype=torch.bool, device=points.device)
    else:
        vis = f["visibility"]
        vis_list = [f["points"][firsts[i]:firsts[i] + lens[i]][vis[firsts[i]:firsts[i] + lens[i]]] for i in range(N)]
        mask = torch.zeros((N * P,), dtype=torch.bool, device=points.device)
        mask[f["flags"].reshape(-1).bool()] = vis
        mask = mask.view(N, P)
    return (vis_list, mask) if return_mask else vis_list


def get_tensor_values(tensor, p, grid_sample=True, mode="bilinear", with_mask=False, squeeze_channel_dim=False):
    """Image values at projected points (DSS/utils/__init__.py:325-375): tensor (B,C,H,W), p (B,N,2)
    in [-1,1] -> (B,N,C) [(B,N) with squeeze_channel_dim], optionally with the finite-value mask.
    `grid_sample=True` is grid_sample(..., mode, padding_mode='reflection') in one kernel
    (iso_image_sample) that writes (B,N,C) directly; `grid_sample=False` is the integer indexing of
    :357-363 (the reference also overwrites the caller's `p` with the pixel coordinates there --
    that side effect is not reproduced)."""
    if not tensor.is_cuda:
        raise RuntimeError("iso_points_amd: the image must be on the GPU; there is no CPU path")
    if mode not in ("bilinear", "nearest"):
        raise ValueError("get_tensor_values: mode must be 'bilinear' or 'nearest', got %r" % (mode,))
    B, C, H, W = tensor.shape
    if not grid_sample:
        assert H == W                                               # :358
    if grid_sample and (tensor.requires_grad or p.requires_grad):
        # a caller that differentiates through the lookup gets the reference's own differentiable op (on the GPU)
        out = torch.nn.functional.grid_sample(tensor, p.unsqueeze(1), mode=mode, padding_mode="reflection",
                                              align_corners=False).squeeze(2).permute(0, 2, 1)
        keep = torch.isfinite(out) if with_mask else None
        if squeeze_channel_dim:
            out, keep = out.squeeze(-1), (keep.squeeze(-1) if with_mask else None)
        return (out, keep) if with_mask else out
    img = tensor.detach().float().contiguous()
    pts = p.detach().float().contiguous()
    assert pts.shape[0] == B and pts.shape[-1] == 2 and pts.dim() == 3
    N = pts.shape[1]
    values = torch.empty((B, N, C), dtype=torch.float32, device=img.device)
    _lib.call("iso_image_sample", _lib.ptr(img), B, C, H, W, _lib.ptr(pts), N,
              (0 if mode == "bilinear" else 1) if grid_sample else 2, _lib.ptr(values), _lib.stream())
    if not grid_sample and bool(torch.isnan(values).any()) and not bool(torch.isnan(img).any()):
        raise IndexError("get_tensor_values: index out of the image (grid_sample=False needs p in [-1,1])")
    mask = torch.isfinite(values) if with_mask else None
    if squeeze_channel_dim:
        values = values.squeeze(-1)
        mask = mask.squeeze(-1) if with_mask else None
    return (values, mask) if with_mask else values
