"""CPU oracle for the surface-splatting half of the hot path -- TEST INFRASTRUCTURE ONLY
(same import rules as iso_oracle.py).

  * raster forward / occupancy backward / zbuf backward: ctypes over oracle/liboracle_splat.so
    (oracle_splat.c), pinned bit-for-bit against the reference's rasterize_points_cpu.cpp
    compiled as-is into oracle/_ref/libdss_ref_cpu.so (`ref_*` functions below).
  * per-point EWA setup, compositing weights, visible-set / backward driver: float32 torch
    restatements of DSS/core/rasterizer.py and DSS/core/renderer.py, pinned through
    tests/golden (make_golden_splat.py runs the reference's methods with shims).
"""
import ctypes
import math
import os
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from .iso_oracle import eps_denom, eps_sqrt, frnn_grid_points

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_splat.so")
_REF_SO = os.path.join(_HERE, "_ref", "libdss_ref_cpu.so")

PointFragments = namedtuple("PointFragments", "idx zbuf qvalue scaler occupancy")

_f = ctypes.POINTER(ctypes.c_float)
_i32 = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.POINTER(ctypes.c_int64)
_u8 = ctypes.POINTER(ctypes.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _np(t, dt):
    return np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=dt)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError("oracle: build liboracle_splat.so first (make -C oracle)")
        _lib = ctypes.CDLL(_SO)
    return _lib


_ref = None


def ref_available():
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (libtorch must be loaded first)
        _ref = ctypes.CDLL(_REF_SO)
    return _ref


# --------------------------------------------------------------------------- raster
def splat_forward(points, ellipse, cutoff, radii, first_idx, num_pts, depth_thres, S, K,
                  bbox_or=True, use_ref=False):
    """RasterizePointsNaive{Cpu,Cuda} semantics (see oracle_splat.c).  Returns torch tensors
    idx i32 (N,S,S,K), zbuf, qvalue f32 (N,S,S,K), occupancy f32 (N,S,S)."""
    pts, el, cu, ra = _np(points, np.float32), _np(ellipse, np.float32), _np(cutoff, np.float32), _np(radii, np.float32)
    fi, nu = _np(first_idx, np.int64), _np(num_pts, np.int64)
    N, P = len(fi), pts.shape[0]
    idx = np.empty((N, S, S, K), np.int32)
    zb = np.empty((N, S, S, K), np.float32)
    qv = np.empty((N, S, S, K), np.float32)
    oc = np.empty((N, S, S), np.float32)
    if use_ref:
        rc = ref().ref_splat_forward(_p(pts, _f), _p(el, _f), _p(cu, _f), _p(ra, _f), _p(fi, _i64), _p(nu, _i64),
                                     ctypes.c_int64(P), N, ctypes.c_float(depth_thres), S, K,
                                     _p(idx, _i32), _p(zb, _f), _p(qv, _f), _p(oc, _f))
        assert rc == 0
    else:
        lib().oracle_splat_forward(_p(pts, _f), _p(el, _f), _p(cu, _f), _p(ra, _f), _p(fi, _i64), _p(nu, _i64),
                                   N, ctypes.c_float(depth_thres), S, K, int(bool(bbox_or)),
                                   _p(idx, _i32), _p(zb, _f), _p(qv, _f), _p(oc, _f))
    return torch.from_numpy(idx), torch.from_numpy(zb), torch.from_numpy(qv), torch.from_numpy(oc)


def occ_backward(points, radii, grad_occ, first_idx, num_pts, radii_s=10.0, rs=None, visible=None,
                 mode=2, use_ref=False):
    """mode 0 = CPU rect, 1 = CUDA rect, 2 = fast disc (default training path).  -> (P,2)."""
    pts, ra, go = _np(points, np.float32), _np(radii, np.float32), _np(grad_occ, np.float32)
    fi, nu = _np(first_idx, np.int64), _np(num_pts, np.int64)
    N, S, P = go.shape[0], go.shape[1], pts.shape[0]
    out = np.zeros((P, 2), np.float32)
    if use_ref:
        assert mode == 0
        rc = ref().ref_occ_backward(_p(pts, _f), _p(ra, _f), _p(go, _f), _p(fi, _i64), _p(nu, _i64),
                                    ctypes.c_int64(P), N, S, ctypes.c_float(radii_s), ctypes.c_float(0.05),
                                    _p(out, _f))
        assert rc == 0
    else:
        rsn = _np(rs, np.float32) if rs is not None else np.zeros((N,), np.float32)
        vis = _np(visible, np.uint8) if visible is not None else None
        lib().oracle_occ_backward(_p(pts, _f), _p(ra, _f), _p(go, _f), _p(fi, _i64), _p(nu, _i64), N, S,
                                  ctypes.c_float(radii_s), _p(rsn, _f), _p(vis, _u8), int(mode), _p(out, _f))
    return torch.from_numpy(out)


def zbuf_backward(idx, grad_zbuf, P, use_ref=False):
    ii, gz = _np(idx, np.int32), _np(grad_zbuf, np.float32)
    N, S, _, K = ii.shape
    out = np.zeros((P,), np.float32)
    if use_ref:
        rc = ref().ref_zbuf_backward(_p(ii, _i32), _p(gz, _f), ctypes.c_int64(P), N, S, K, _p(out, _f))
        assert rc == 0
    else:
        lib().oracle_zbuf_backward(_p(ii, _i32), _p(gz, _f), N, S, K, _p(out, _f))
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- cameras (synthetic)
def look_at_view(dist, elev_deg, azim_deg):
    """World->view 4x4 in pytorch3d's row-vector convention (p_view = [p,1] @ V):
    camera at spherical (dist, elev, azim) looking at the origin, +Y up; view axes
    +X left, +Y up, +Z into the scene (pytorch3d look_at_view_transform)."""
    e, a = math.radians(elev_deg), math.radians(azim_deg)
    C = torch.tensor([dist * math.cos(e) * math.sin(a), dist * math.sin(e), dist * math.cos(e) * math.cos(a)])
    z = F.normalize(-C, dim=0)
    x = F.normalize(torch.cross(torch.tensor([0.0, 1.0, 0.0]), z, dim=0), dim=0)
    y = F.normalize(torch.cross(z, x, dim=0), dim=0)
    R = torch.stack([x, y, z], dim=1)          # columns = view axes in world coords
    T = -(C @ R)
    V = torch.eye(4)
    V[:3, :3] = R
    V[3, :3] = T
    return V


def perspective(fov_deg, znear=1.0, zfar=100.0):
    """View->NDC 4x4, row-vector convention (pytorch3d FoVPerspectiveCameras, aspect 1)."""
    t = math.tan(math.radians(fov_deg) / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / t
    P[1, 1] = 1 / t
    P[2, 2] = zfar / (zfar - znear)
    P[3, 2] = -(zfar * znear) / (zfar - znear)
    P[2, 3] = 1.0
    return P


# --------------------------------------------------------------------------- per-point EWA setup
def filter_renderable(points, normals, V, znear=1.0, zfar=100.0, backface_culling=True):
    """SurfaceSplatting.filter_renderable for one view (rasterizer.py:184-254):
    znear <= z_view <= zfar and (view-space normal).z < 0."""
    ph = torch.cat([points, torch.ones_like(points[:, :1])], dim=-1)
    zv = (ph @ V)[:, 2]
    mask = (zv >= znear) & (zv <= zfar)
    if backface_culling:
        nz = (normals @ V[:3, :3])[:, 2]
        mask = mask & (nz < 0)
    return mask


def vrk_h(points_padded, num_points, frnn_radius=0.2, frnn_fn=None):
    """rasterizer.py:367-386: h = clamp(0.5*max_{6 nn} d2, 5e-5, 0.01); clouds with < 7 points
    get d2 = 1e-3.  Returns packed (sum P,)."""
    frnn_fn = frnn_fn or frnn_grid_points
    sq, _, _, _ = frnn_fn(points_padded, points_padded, num_points, num_points, K=7, r=frnn_radius)
    sq = sq[:, :, 1:].clone()
    sq[num_points < 7] = 1e-3
    packed = torch.cat([sq[b, : int(n)] for b, n in enumerate(num_points.tolist())], dim=0)
    h = 0.5 * packed.max(dim=-1, keepdim=True)[0]
    return h.clamp(5e-5, 0.01).view(-1)


def tangent_frame(normals):
    """Deterministic instance of the reference's random tangent frame (rasterizer.py:395-397:
    u0 = normalize(n x (n + rand)), u1 = normalize(n x u0)): the helper vector is the
    coordinate axis least aligned with n.  Sk^T Sk = I - n n^T / |n|^2 for any valid choice."""
    a = normals.abs()
    e = torch.zeros_like(normals)
    k = a.argmin(dim=-1)
    e[torch.arange(normals.shape[0]), k] = 1.0
    u0 = F.normalize(torch.cross(normals, normals + e, dim=-1), dim=-1)
    u1 = F.normalize(torch.cross(normals, u0, dim=-1), dim=-1)
    return torch.stack([u0, u1], dim=1)


def per_point_info(points, normals, h, M44, image_size, cutoff=1.0, sigma=1.0, Sk=None, dtype=None):
    """SurfaceSplatting._get_per_point_info for ONE view (rasterizer.py:441-563).
    points/normals (P,3) already filtered, h (P,), M44 full world->NDC projection (4,4).
    dtype=torch.float64 evaluates the same formulas in double (ground truth for tests)."""
    if dtype is not None:
        points, normals, h, M44 = points.to(dtype), normals.to(dtype), h.to(dtype), M44.to(dtype)
    P = points.shape[0]
    ph = torch.cat([points, torch.ones_like(points[:, :1])], dim=-1)          # to_homogen
    W = M44[:3, :].expand(P, 3, 4)
    denom = (ph[:, None, :] @ M44[:, 3:].expand(P, 4, 1)).view(-1)             # :466-468
    denom_sqr = eps_denom(denom ** 2)
    Jk = ph.new_zeros(P, 4, 2)
    denom = eps_denom(denom)
    Jk[:, 0, 0] = 1 / denom
    Jk[:, 1, 1] = 1 / denom
    xy = ph[:, None, :] @ M44[:, :2].expand(P, 4, 2)                           # (P,1,2)
    Jk[:, 3, 0] = -1 / denom_sqr * xy[:, :, 0].view(-1)
    Jk[:, 3, 1] = -1 / denom_sqr * xy[:, :, 1].view(-1)
    WJk = W @ Jk                                                              # (P,3,2)
    if Sk is None:
        Sk = tangent_frame(normals)
    Vrk = h.view(-1, 1, 1) * Sk.transpose(1, 2) @ Sk
    Mk = Sk @ WJk
    Vk = WJk.transpose(1, 2) @ Vrk @ WJk
    pixel_size = 2.0 / image_size
    GV = Vk + sigma * torch.eye(2, dtype=points.dtype).expand(P, 2, 2) * (pixel_size ** 2)
    detMk = torch.det(Mk)
    GVdet = torch.det(GV)
    GVinv = torch.inverse(GV)
    ellipse = torch.stack([GVinv[:, 0, 0], GVinv[:, 0, 1] + GVinv[:, 1, 0], GVinv[:, 1, 1]], dim=-1)
    a, b, c = ellipse[:, 0], ellipse[:, 1], ellipse[:, 2]
    den = eps_denom(4 * a * c - b ** 2)
    y = torch.sqrt(eps_sqrt(4 * a * cutoff / den))
    x = torch.sqrt(eps_sqrt(4 * c * cutoff / den))
    radii = torch.stack([x, y], dim=-1)
    scaler = torch.sqrt(eps_sqrt(GVdet * 4 * np.pi * np.pi))
    scaler = detMk.abs() / eps_denom(scaler)
    return {"radii": radii, "ellipse_params": ellipse, "cutoff_threshold": torch.full_like(a, cutoff),
            "scaler": scaler}


def transform_to_ndc(points, V, M44):
    """PointsRasterizer.transform (pytorch3d): xy = full projection / w, z = view-space depth."""
    ph = torch.cat([points, torch.ones_like(points[:, :1])], dim=-1)
    o = ph @ M44
    ndc = o[:, :3] / o[:, 3:]
    ndc[:, 2] = (ph @ V)[:, 2]
    return ndc


# --------------------------------------------------------------------------- compositing
def composite(fragments, features, norm_weighted=True, eps=1e-4):
    """SurfaceSplattingRenderer.forward (renderer.py:53-78): w = exp(-0.5 q) * scaler,
    rgb = sum w f / max(sum w, eps) (pytorch3d NormWeightedCompositor) or sum w f;
    alpha = occupancy.  features (P,C) packed -> (N,S,S,C+1)."""
    idx = fragments.idx.long()
    w = torch.exp(-0.5 * fragments.qvalue) * fragments.scaler
    w = torch.where(idx >= 0, w, torch.zeros_like(w))
    f = features[idx.clamp(min=0)] * (idx >= 0)[..., None].to(features.dtype)
    num = (w[..., None] * f).sum(dim=-2)
    if norm_weighted:
        num = num / w.sum(dim=-1, keepdim=True).clamp(min=eps)
    return torch.cat([num, fragments.occupancy[..., None]], dim=-1)


def gather_scaler(scaler, idx):
    """gather_with_neg_idx (utils/__init__.py:172-190): scaler[idx], 0 where idx < 0."""
    s = scaler[idx.long().clamp(min=0)]
    return torch.where(idx >= 0, s, torch.zeros_like(s))


# --------------------------------------------------------------------------- backward driver
def lower_median(x):
    """torch.median of a flattened tensor: the lower of the two middle elements."""
    return torch.sort(x.reshape(-1))[0][(x.numel() - 1) // 2]


def splat_backward(points, radii, idx, first_idx, num_pts, occ_grad, zbuf_grad, radii_s=10.0):
    """EllipticalRasterizer.backward, default fast path (rasterizer.py:841-968):
    visible = points listed in pixels whose first slot is filled; r_n = median(radii of the
    visible points of cloud n, both columns) * radii_s; disc-support occupancy gradient for
    visible points + zbuf gradient scatter.  Returns (P,3)."""
    P = points.shape[0]
    mask = idx[..., 0] >= 0
    vis = torch.zeros(P, dtype=torch.bool)
    sel = idx[mask].reshape(-1).long()
    vis[sel[sel >= 0]] = True
    rs = []
    for n in range(len(first_idx)):
        s, e = int(first_idx[n]), int(first_idx[n]) + int(num_pts[n])
        rv = radii[s:e][vis[s:e]]
        rs.append((lower_median(rv) * radii_s).item() if rv.numel() else 0.0)
    rs = torch.tensor(rs, dtype=torch.float32)
    gxy = occ_backward(points, radii, occ_grad, first_idx, num_pts, radii_s, rs=rs,
                       visible=vis.to(torch.uint8), mode=2)
    gz = zbuf_backward(idx, zbuf_grad, P)
    return torch.cat([gxy, gz[:, None]], dim=-1), vis, rs
