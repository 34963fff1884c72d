"""Synthetic camera matrices for tests / bench (pytorch3d conventions: row vectors,
p' = [p,1] @ M; view space +X left, +Y up, +Z into the scene).  Cameras themselves are
out of scope: the splat operators only take the two 4x4 matrices."""
import math

import torch
import torch.nn.functional as F


def look_at_view(dist, elev_deg, azim_deg):
    """World->view matrix of a camera at spherical (dist, elev, azim) looking at the origin."""
    e, a = math.radians(elev_deg), math.radians(azim_deg)
    C = torch.tensor([dist * math.cos(e) * math.sin(a), dist * math.sin(e), dist * math.cos(e) * math.cos(a)])
    z = F.normalize(-C, dim=0)
    x = F.normalize(torch.cross(torch.tensor([0.0, 1.0, 0.0]), z, dim=0), dim=0)
    y = F.normalize(torch.cross(z, x, dim=0), dim=0)
    R = torch.stack([x, y, z], dim=1)
    V = torch.eye(4)
    V[:3, :3] = R
    V[3, :3] = -(C @ R)
    return V


def perspective(fov_deg, znear=1.0, zfar=100.0):
    """View->NDC matrix (field of view in degrees, aspect 1)."""
    t = math.tan(math.radians(fov_deg) / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / t
    P[1, 1] = 1 / t
    P[2, 2] = zfar / (zfar - znear)
    P[3, 2] = -(zfar * znear) / (zfar - znear)
    P[2, 3] = 1.0
    return P
