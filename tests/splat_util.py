"""Synthetic splat scenes shared by the splat parity tests (CPU and GPU)."""
import math

import torch

from oracle import splat_oracle as SO


def sphere_scene(P, n_views=2, S=64, seed=0, dist=3.0, fov=30.0, frnn_radius=0.2, radius=1.0):
    """Unit-sphere samples + outward normals seen from `n_views` look-at cameras
    (SURVEY 8(d) cfg 3 at test size).  Returns per-view packed, filtered splat inputs exactly
    as SurfaceSplatting.forward would hand them to rasterize_elliptical_points."""
    g = torch.Generator().manual_seed(seed)
    pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1) * radius
    nrm = torch.nn.functional.normalize(pts + 0.05 * torch.randn(P, 3, generator=g), dim=-1)
    Vs = [SO.look_at_view(dist, 20.0, 360.0 * i / n_views) for i in range(n_views)]
    Pm = SO.perspective(fov)
    packed = {"points": [], "normals": [], "ndc": [], "ellipse": [], "cutoff": [], "radii": [], "scaler": []}
    num = []
    keep = []
    for V in Vs:
        m = SO.filter_renderable(pts, nrm, V)
        keep.append(m)
        p, n = pts[m], nrm[m]
        num.append(p.shape[0])
        packed["points"].append(p)
        packed["normals"].append(n)
    num_t = torch.tensor(num)
    mx = max(num)
    padded = torch.zeros(n_views, mx, 3)
    for i, p in enumerate(packed["points"]):
        padded[i, : num[i]] = p
    h = SO.vrk_h(padded, num_t, frnn_radius)
    s = 0
    for i, V in enumerate(Vs):
        M44 = V @ Pm
        info = SO.per_point_info(packed["points"][i], packed["normals"][i], h[s : s + num[i]], M44, S)
        s += num[i]
        packed["ndc"].append(SO.transform_to_ndc(packed["points"][i], V, M44))
        packed["ellipse"].append(info["ellipse_params"])
        packed["cutoff"].append(info["cutoff_threshold"])
        packed["radii"].append(info["radii"])
        packed["scaler"].append(info["scaler"])
    out = {k: torch.cat(v, 0).contiguous() for k, v in packed.items()}
    out["num"] = num_t
    out["first"] = torch.cumsum(num_t, 0) - num_t
    out["h"] = h
    out["views"] = torch.stack(Vs)
    out["proj"] = Pm
    out["world_points"], out["world_normals"], out["keep"] = pts, nrm, torch.stack(keep)
    out["S"] = S
    return out


def random_splats(P, N=2, seed=0, pad=1.0):
    """Unstructured splats: random NDC positions (some behind the camera, some off screen),
    random positive-definite ellipses, radii = true bbox * pad."""
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(P, 3, generator=g) * torch.tensor([2.4, 2.4, 3.0]) - torch.tensor([1.2, 1.2, 0.3])
    sx = torch.rand(P, generator=g) * 0.08 + 0.01
    sy = torch.rand(P, generator=g) * 0.08 + 0.01
    rho = (torch.rand(P, generator=g) - 0.5) * 1.6
    # inverse covariance of a rotated gaussian
    den = (1 - rho ** 2)
    a = 1 / (sx ** 2 * den)
    c = 1 / (sy ** 2 * den)
    b = -2 * rho / (sx * sy * den)
    cutoff = torch.rand(P, generator=g) * 1.5 + 0.5
    d = 4 * a * c - b ** 2
    rx = torch.sqrt(4 * c * cutoff / d) * pad
    ry = torch.sqrt(4 * a * cutoff / d) * pad
    # a few exact depth ties to exercise the (z, idx) rule
    pts[::17, 2] = 1.25
    num = torch.tensor([P // N] * (N - 1) + [P - (P // N) * (N - 1)])
    return {"ndc": pts.contiguous(), "ellipse": torch.stack([a, b, c], -1).contiguous(),
            "cutoff": cutoff.contiguous(), "radii": torch.stack([rx, ry], -1).contiguous(),
            "num": num, "first": torch.cumsum(num, 0) - num}
