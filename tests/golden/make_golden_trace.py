"""Golden vectors for the ray queries, made by the REFERENCE's own
SphereTracing.project_points and find_zero_crossing_between_point_pairs
(DSS/models/levelset_sampling.py:663-808, :1210-1367) imported through make_golden.py's shims.
usage:  ISO_GOLDEN_ONLY=trace python tests/golden/make_golden.py"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import torch

from oracle import iso_oracle as O
from make_golden import npz, siren_arrays


def camera_rays(n, seed, cam=(0.0, 0.3, 2.5), spread=0.55):
    """Rays from one camera toward a disc around the origin, started on the sphere of radius 1.05
    (where pixels_to_world, implicit_modeling.py:296-306, starts them: the bounding-volume entry)."""
    g = torch.Generator().manual_seed(seed)
    c = torch.tensor(cam)
    target = (torch.rand(n, 3, generator=g) - 0.5) * 2 * spread
    d = torch.nn.functional.normalize(target - c, dim=-1)
    # entry point of the ray into the sphere |x| = 1.05 (all of these rays hit it)
    b = (d * c).sum(-1)
    disc = b * b - (c.dot(c) - 1.05 ** 2)
    t = -b - torch.sqrt(disc.clamp_min(0))
    return (c + t[:, None] * d).view(1, n, 3), d.view(1, n, 3)


class _WithC(torch.nn.Module):
    """The reference passes c=... to the decoder; the oracle models take **kwargs."""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x, c=None, **kw):
        return self.m.forward(x)


def gen_trace(L):
    ST = L.SphereTracing
    sph = O.SphereSDF((0.05, -0.1, 0.0), 0.6)
    r0, d = camera_rays(3000, 5)
    out = ST(proj_max_iters=10).project_points(r0.clone(), d.clone(), _WithC(sph))
    out3 = ST(proj_max_iters=3, alpha=0.8).project_points(r0.clone(), d.clone(), _WithC(sph))
    npz("trace_sphere.npz", ray0=r0, dirs=d, center=[0.05, -0.1, 0.0], radius=0.6,
        T10_points=out["levelset_points"], T10_eval=out["network_eval_on_levelset_points"], T10_mask=out["mask"],
        T3_points=out3["levelset_points"], T3_eval=out3["network_eval_on_levelset_points"], T3_mask=out3["mask"])
    torch.manual_seed(0)
    m_fit = O.fit_siren_to_sphere(O.SirenSDF(hidden_size=256, n_layers=3), radius=0.7, steps=300)
    r0, d = camera_rays(2000, 6, spread=0.9)
    out = ST(proj_max_iters=10).project_points(r0.clone(), d.clone(), _WithC(m_fit))
    # zero crossing between the front hit and the far side of the bounding sphere
    far = r0 + d * 2.0
    zc, zmask = L.find_zero_crossing_between_point_pairs(r0.clone(), far.clone(), _WithC(m_fit), is_occupancy=False)
    zc_s, zmask_s = L.find_zero_crossing_between_point_pairs(r0[:, :500].clone(), far[:, :500].clone(), _WithC(sph),
                                                           is_occupancy=False, n_steps=64, n_secant_steps=6)
    npz("trace_siren.npz", ray0=r0, dirs=d, T10_points=out["levelset_points"],
        T10_eval=out["network_eval_on_levelset_points"], T10_mask=out["mask"],
        zc_p1=far, zc_points=zc, zc_mask=zmask, zc_sphere_points=zc_s, zc_sphere_mask=zmask_s,
        center=[0.05, -0.1, 0.0], radius=0.6, **siren_arrays(m_fit))
    torch.manual_seed(2)
    idr = O.IdrSDF(hidden_size=128, n_layers=4, num_frequencies=4, skip_in=(2,))
    r0, d = camera_rays(1500, 7, spread=0.9)
    out = ST(proj_max_iters=12).project_points(r0.clone(), d.clone(), _WithC(idr))
    npz("trace_idr.npz", ray0=r0, dirs=d, T=12, out_points=out["levelset_points"],
        out_eval=out["network_eval_on_levelset_points"], out_mask=out["mask"], idr_raw=idr.raw_weights(),
        idr_hidden=128, idr_layers=4, idr_freq=4, idr_skip=2)

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_trace.py
    _os.environ["ISO_GOLDEN_ONLY"] = "trace"
    import make_golden
    make_golden.main()
