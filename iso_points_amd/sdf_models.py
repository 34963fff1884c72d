"""SDF network definitions the fused projection kernels understand, and the weight
extraction that feeds them.

`Siren` / `SineLayer` mirror the *forward definition* of the reference's classes
(DSS/models/common.py:56-165: same constructor arguments, same attribute layout
`net[i].linear.{weight,bias}` / `omega_0`, same initialisation) so that a model built
by the reference's config factory and one built here are interchangeable for
`UniformProjection`.  Weights are re-read on every call (they change every optimiser
step).
"""
from collections import OrderedDict, namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib

_fields = ("sdf", "latent", "rgb", "occupancy")
NetOutput = namedtuple("Result", _fields, defaults=(None,) * len(_fields))


class SphereSDF(nn.Module):
    """Analytic |x-c|-R (BASELINE.json configs[0]); `iso_analytic` routes it to
    iso_project_sphere."""
    iso_analytic = "sphere"

    def __init__(self, center=(0.0, 0.0, 0.0), radius=1.0):
        super().__init__()
        self.register_buffer("center", torch.tensor(center, dtype=torch.float32))
        self.radius = float(radius)

    def forward(self, x, **kwargs):
        return NetOutput(sdf=(x - self.center).norm(dim=-1, keepdim=True) - self.radius)


class SineLayer(nn.Module):
    """common.py:56-87."""

    def __init__(self, dim, out_dim, bias=True, is_first=False, omega_0=30):
        super().__init__()
        self.omega_0 = omega_0
        self.is_first = is_first
        self.dim = dim
        self.linear = nn.Linear(dim, out_dim, bias=bias)
        with torch.no_grad():
            if is_first:
                self.linear.weight.uniform_(-1 / dim, 1 / dim)
            else:
                b = np.sqrt(6 / dim) / omega_0
                self.linear.weight.uniform_(-b, b)

    def forward(self, input):
        return torch.sin(self.omega_0 * self.linear(input))


class Siren(nn.Module):
    """common.py:90-165 restricted to what the hot path uses: sdf output, linear head,
    no latent code (c_dim=0 as in test_dtu_points.py:216-227)."""

    def __init__(self, dim=3, hidden_size=256, n_layers=3, out_dims=None, outermost_linear=True,
                 c_dim=0, first_omega_0=30, hidden_omega_0=30.0, activation=None, **kwargs):
        super().__init__()
        out_dims = out_dims or OrderedDict(sdf=1)
        if c_dim != 0 or not outermost_linear or activation is not None or sum(out_dims.values()) != 1:
            raise NotImplementedError("iso_points_amd.Siren covers the sdf-only, c_dim=0, linear-head "
                                      "configuration of the reference's Siren")
        self.dim, self.c_dim = dim, c_dim
        net = [SineLayer(dim, hidden_size, is_first=True, omega_0=first_omega_0)]
        for _ in range(n_layers):
            net.append(SineLayer(hidden_size, hidden_size, is_first=False, omega_0=hidden_omega_0))
        final = nn.Linear(hidden_size, 1)
        with torch.no_grad():
            b = np.sqrt(6 / hidden_size) / hidden_omega_0
            final.weight.uniform_(-b, b)
        net.append(final)
        self.net = nn.Sequential(*net)

    def forward(self, coords, c=None, **kwargs):
        return NetOutput(sdf=self.net(coords))


def siren_spec(model):
    """Return (linears, omega_first, omega_hidden) if `model` is a SIREN the fused kernel
    can run (3 -> H -> H.. -> 1, sine layers with equal hidden omega, linear head), else None.
    Accepts the reference's Siren (`.net`), ours, and the oracle's SirenSDF (`.lins`)."""
    lins, omegas = None, None
    if hasattr(model, "net") and isinstance(model.net, nn.Sequential):
        mods = list(model.net)
        if len(mods) < 2 or not isinstance(mods[-1], nn.Linear):
            return None
        if getattr(model, "c_dim", 0) not in (0, None):
            return None
        if not all(hasattr(m, "linear") and hasattr(m, "omega_0") for m in mods[:-1]):
            return None
        lins = [m.linear for m in mods[:-1]] + [mods[-1]]
        omegas = [float(m.omega_0) for m in mods[:-1]]
    elif hasattr(model, "lins") and hasattr(model, "first_omega_0"):
        lins = list(model.lins)
        omegas = [float(model.first_omega_0)] + [float(model.hidden_omega_0)] * (len(lins) - 2)
    else:
        return None
    H = lins[0].out_features
    if lins[0].in_features != 3 or lins[-1].out_features != 1 or lins[-1].in_features != H:
        return None
    if H not in (64, 128, 256) or len(lins) - 2 > 8:
        return None
    for lin in lins[1:-1]:
        if lin.in_features != H or lin.out_features != H:
            return None
    if any(lin.bias is None for lin in lins):
        return None
    if len(set(omegas[1:])) > 1:
        return None
    return lins, omegas[0], (omegas[1] if len(omegas) > 1 else omegas[0])


class PackedSiren(object):
    """Device-side MFMA weight image of a SIREN (iso_siren_pack_weights)."""

    def __init__(self, model, device):
        spec = siren_spec(model)
        if spec is None:
            raise ValueError("model is not a SIREN the fused kernel supports")
        lins, self.omega_first, self.omega_hidden = spec
        self.hidden = lins[0].out_features
        self.n_hidden = len(lins) - 2
        parts = []
        for lin in lins:
            parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        raw = torch.cat(parts).to(device=device, dtype=torch.float32).contiguous()
        lib = _lib.load()
        assert raw.numel() == lib.iso_siren_raw_floats(self.hidden, self.n_hidden)
        self.packed = torch.empty((lib.iso_siren_packed_floats(self.hidden, self.n_hidden),),
                                  dtype=torch.float32, device=device)
        _lib.call("iso_siren_pack_weights", _lib.ptr(raw), _lib.ptr(self.packed), self.hidden,
                  self.n_hidden, _lib.stream())
        self._ws = None

    def workspace(self, n):
        need = _lib.load().iso_project_siren_workspace_bytes(int(n), self.hidden, self.n_hidden)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.packed.device)
        return self._ws


def siren_sdf_and_grad(model, points):
    """One fused SDF + gradient evaluation (UniformProjection._compute_sdf_and_grad,
    levelset_sampling.py:142-170) for a SIREN.  points (...,3) -> sdf (...), grad (...,3)."""
    shp = points.shape
    pts = points.detach().reshape(-1, 3).float().contiguous()
    ps = PackedSiren(model, pts.device)
    n = pts.shape[0]
    sdf = torch.empty((n,), dtype=torch.float32, device=pts.device)
    grad = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
    ws = ps.workspace(n)
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), n,
              _lib.ptr(ps.packed), ps.hidden, ps.n_hidden, ps.omega_first, ps.omega_hidden,
              _lib.ptr(ws), ws.numel(), _lib.stream())
    return sdf.view(shp[:-1]), grad.view(shp)
