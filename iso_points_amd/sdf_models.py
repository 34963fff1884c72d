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
    if H < 1 or H > 256 or len(lins) - 2 > 8:       # widths other than 64 / 128 / 256 are zero-padded up (PackedSiren)
        return None
    for lin in lins[1:-1]:
        if lin.in_features != H or lin.out_features != H:
            return None
    if any(lin.bias is None for lin in lins):
        return None
    if len(set(omegas[1:])) > 1:
        return None
    return lins, omegas[0], (omegas[1] if len(omegas) > 1 else omegas[0])


def weights_key(lins, device):
    """Identity of a network's weights as the packed images see them: storage address and in-place version of every
    weight / bias.  An optimiser step (in-place) or a re-assigned .data changes it -- but an in-place update THROUGH
    `.data` (p.data.mul_(), p.data.copy_(): EMA, weight clipping, hand-written optimisers) or a raw-pointer write bumps
    neither, so this key is only a sanity check inside one operator call (UniformProjection._packing); across calls a
    weight image is re-used on the caller's explicit word only (reuse_packed)."""
    return (str(device),) + tuple((t.data_ptr(), t._version) for lin in lins for t in (lin.weight, lin.bias))


class PackedSiren(object):
    """Device-side MFMA weight image of a SIREN (iso_siren_pack_weights).  `key` = weights_key at packing time;
    `current(model)` is necessary, not sufficient, for the image to be up to date (see weights_key)."""

    def current(self, model, device):
        spec = siren_spec(model)
        return spec is not None and weights_key(spec[0], device) == self.key

    def __init__(self, model, device):
        spec = siren_spec(model)
        if spec is None:
            raise ValueError("model is not a SIREN the fused kernel supports")
        lins, self.omega_first, self.omega_hidden = spec
        self.key = weights_key(lins, device)
        self.model_hidden = lins[0].out_features
        # the fused kernels exist for H = 64 / 128 / 256: any other width runs as the next one up with ZERO rows and
        # columns added -- a padded unit has z = 0, sin(0) = 0 and feeds zero columns, so value and gradient are those
        # of the unpadded network (zeros add exactly nothing to a sum)
        self.hidden = H = next(h for h in (64, 128, 256) if h >= self.model_hidden)
        self.n_hidden = len(lins) - 2
        parts = []
        for i, lin in enumerate(lins):
            w = lin.weight.detach().to(device=device, dtype=torch.float32)
            b = lin.bias.detach().to(device=device, dtype=torch.float32)
            rows = H if i < len(lins) - 1 else w.shape[0]
            cols = H if i > 0 else w.shape[1]
            if (rows, cols) != tuple(w.shape):
                wp = w.new_zeros((rows, cols))
                wp[:w.shape[0], :w.shape[1]] = w
                bp = b.new_zeros((rows,))
                bp[:b.shape[0]] = b
                w, b = wp, bp
            parts += [w.reshape(-1), b.reshape(-1)]
        raw = torch.cat(parts).contiguous()
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


def siren_sdf_and_grad(model, points, need_grad=True, packed=None):
    """One fused SDF + gradient evaluation (UniformProjection._compute_sdf_and_grad,
    levelset_sampling.py:142-170) for a SIREN.  points (...,3) -> sdf (...), grad (...,3).
    need_grad=False: value only (forward sweep only), grad is None."""
    shp = points.shape
    pts = points.detach().reshape(-1, 3).float().contiguous()
    ps = packed if packed is not None else PackedSiren(model, pts.device)
    n = pts.shape[0]
    sdf = torch.empty((n,), dtype=torch.float32, device=pts.device)
    grad = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if need_grad else None
    ws = ps.workspace(n)
    _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), n,
              _lib.ptr(ps.packed), ps.hidden, ps.n_hidden, ps.omega_first, ps.omega_hidden,
              _lib.ptr(ws), ws.numel(), _lib.stream())
    return sdf.view(shp[:-1]), (grad.view(shp) if need_grad else None)


# ----------------------------------------------------------------------------- IDR-style SDF
def _effective_weight(lin):
    """Weight of a (possibly weight-normalised) nn.Linear: g * v / |v| (common.py:277-278)."""
    if hasattr(lin, "weight_g") and hasattr(lin, "weight_v"):
        v = lin.weight_v
        return v * (lin.weight_g / v.norm(dim=1, keepdim=True))
    par = getattr(lin, "parametrizations", None)
    if par is not None and hasattr(par, "weight"):
        return lin.weight                       # new-style parametrization recomputes on access
    return lin.weight


def idr_spec(model):
    """(weights, biases, hidden, n_layers, skip, n_freq) if `model` is an IDR-style SDF
    (DSS/models/common.py:220-310; also the oracle's IdrSDF) the fused kernel supports, else None."""
    try:
        if hasattr(model, "v") and hasattr(model, "g") and hasattr(model, "dims"):      # oracle IdrSDF
            n_lin = model.num_layers - 1
            Ws = [model.weight(l) for l in range(n_lin)]
            bs = [model.b[l] for l in range(n_lin)]
            F_ = model.F
            skip_in = tuple(model.skip_in)
        elif hasattr(model, "lin0") and hasattr(model, "skip_in") and hasattr(model, "num_layers"):
            n_lin = model.num_layers - 1
            lins = [getattr(model, "lin%d" % l) for l in range(n_lin)]
            Ws = [_effective_weight(l) for l in lins]
            bs = [l.bias for l in lins]
            d0 = Ws[0].shape[1]
            if model.embed_fn is None or (d0 - 3) % 6 != 0:
                return None
            F_ = (d0 - 3) // 6
            skip_in = tuple(model.skip_in)
            sp = getattr(model, "softplus", None)
            if sp is None or abs(float(sp.beta) - 100.0) > 0 or float(getattr(sp, "threshold", 20)) != 20:
                return None
        else:
            return None
        n_layers = n_lin - 1
        H = Ws[-1].shape[1]
        if len(skip_in) > 1 or H not in (128, 256, 512) or not (2 <= n_layers <= 12) or F_ > 10:
            return None
        skip = skip_in[0] if len(skip_in) == 1 else -1
        d0 = 3 + 6 * F_
        if Ws[0].shape[1] != d0 or Ws[-1].shape[0] != 1:
            return None
        for l in range(n_layers):
            want = H - d0 if (skip >= 1 and l == skip - 1) else H
            if Ws[l].shape[0] != want or (l > 0 and Ws[l].shape[1] != H):
                return None
        if skip == 0 or skip >= n_layers:
            return None
        return Ws, bs, H, n_layers, skip, F_
    except Exception:
        return None


class PackedIdr(object):
    """Device-side MFMA weight image of an IDR-style SDF (iso_idr_pack_weights)."""

    def __init__(self, model, device):
        spec = idr_spec(model)
        if spec is None:
            raise ValueError("model is not an IDR-style SDF the fused kernel supports")
        Ws, bs, self.hidden, self.n_layers, self.skip, self.n_freq = spec
        parts = []
        for W, b in zip(Ws, bs):
            parts += [W.detach().reshape(-1), b.detach().reshape(-1)]
        raw = torch.cat(parts).to(device=device, dtype=torch.float32).contiguous()
        lib = _lib.load()
        assert raw.numel() == lib.iso_idr_raw_floats(self.hidden, self.n_layers, self.skip, self.n_freq)
        self.packed = torch.empty((lib.iso_idr_packed_floats(self.hidden, self.n_layers),), dtype=torch.float32,
                                  device=device)
        _lib.call("iso_idr_pack_weights", _lib.ptr(raw), _lib.ptr(self.packed), self.hidden, self.n_layers,
                  self.skip, self.n_freq, _lib.stream())
        self._ws = None

    def workspace(self, n):
        need = _lib.load().iso_project_idr_workspace_bytes(int(n), self.hidden, self.n_layers)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.packed.device)
        return self._ws


def idr_sdf_and_grad(model, points, need_grad=True, packed=None):
    shp = points.shape
    pts = points.detach().reshape(-1, 3).float().contiguous()
    pk = packed if packed is not None else PackedIdr(model, pts.device)
    n = pts.shape[0]
    sdf = torch.empty((n,), dtype=torch.float32, device=pts.device)
    grad = torch.empty((n, 3), dtype=torch.float32, device=pts.device) if need_grad else None
    ws = pk.workspace(n)
    _lib.call("iso_idr_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), n, _lib.ptr(pk.packed), pk.hidden,
              pk.n_layers, pk.skip, pk.n_freq, 100.0, _lib.ptr(ws), ws.numel(), _lib.stream())
    return sdf.view(shp[:-1]), (grad.view(shp) if need_grad else None)


class FusedSdf(object):
    """`sdf(points) -> values` for the callers that only need the value: the `sdf` callable of
    RayTracing (levelset_sampling.py:831-1167), the proposal / secant evaluations of
    find_zero_crossing_between_point_pairs (:1262-1267, :1350-1353) and the ray candidates of
    combined_modeling.py:376-380.  The weight image is packed once per instance (build one per
    optimiser step); SIREN and IDR-style networks run forward-only fused kernels, an analytic
    sphere or any other nn.Module is evaluated with model.forward on the GPU."""

    def __init__(self, model, device):
        self.model = model
        self.kind, self.packed = "generic", None
        if siren_spec(model) is not None:
            self.kind, self.packed = "siren", PackedSiren(model, device)
        elif idr_spec(model) is not None:
            self.kind, self.packed = "idr", PackedIdr(model, device)

    def __call__(self, points, **forward_kwargs):
        """points (...,3) -> sdf (...)"""
        if not points.is_cuda:
            raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
        if forward_kwargs or self.kind == "generic":
            with torch.no_grad():
                return self.model.forward(points.reshape(-1, 3), **forward_kwargs).sdf.reshape(points.shape[:-1])
        if self.kind == "siren":
            return siren_sdf_and_grad(self.model, points, need_grad=False, packed=self.packed)[0]
        return idr_sdf_and_grad(self.model, points, need_grad=False, packed=self.packed)[0]
