"""Golden vectors for IDR's two-ended ray tracer, made by the REFERENCE's own
RayTracing.forward (DSS/models/levelset_sampling.py:810-1167) imported through make_golden.py's
shims.  The reference hard-codes `.cuda()` on every temporary; this image has no GPU, so
Tensor.cuda is the identity while the fixtures are made (a run-time shim, no source edit).
usage:  ISO_GOLDEN_ONLY=raytrace python tests/golden/make_golden.py"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import torch

from oracle import iso_oracle as O
from make_golden import npz, siren_arrays


def pixel_rays(n, seed, cam=(0.0, 0.3, 2.5), spread=1.25):
    """Unit rays from one camera toward a square around the origin; with spread 1.25 about a
    third of them miss the unit bounding sphere (the tangent-plane branch, utils/__init__.py:533)."""
    g = torch.Generator().manual_seed(seed)
    c = torch.tensor(cam)
    target = (torch.rand(n, 3, generator=g) - 0.5) * 2 * spread
    return c.view(1, 3), torch.nn.functional.normalize(target - c, dim=-1).view(1, n, 3)


def silhouette(cam, dirs, center, radius):
    """Ground-truth object mask = rays hitting a sphere (center, radius): a shifted copy of the
    traced shape, so all four (in / out of the mask) x (hit / miss) combinations occur."""
    c = torch.tensor(center).view(1, 1, 3)
    oc = cam.view(1, 1, 3) - c
    b = (dirs * oc).sum(-1)
    return ((b * b - ((oc * oc).sum(-1) - radius ** 2)) > 0).view(-1)


def run(L, sdf, cam, dirs, gt, training, seed, **kw):
    rt = L.RayTracing(**kw)
    rt.train(training)
    torch.manual_seed(seed)
    u = torch.empty(rt.n_steps).uniform_(0.0, 1.0)     # the draw of :1142 under this seed
    torch.manual_seed(seed)
    with torch.no_grad():
        pts, mask, z = rt(sdf=sdf, cam_loc=cam.clone(), object_mask=gt.clone(), ray_directions=dirs.clone())
    return {"points": pts, "mask": mask, "dist": z, "uniform": u}


def gen_raytrace(L):
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        sph = O.SphereSDF((0.05, -0.1, 0.0), 0.6)
        f_sph = lambda x: sph.forward(x).sdf.reshape(-1)
        cam, dirs = pixel_rays(2500, 11)
        gt = silhouette(cam, dirs, (0.1, -0.05, 0.0), 0.62)
        out = {"cam": cam, "dirs": dirs, "gt": gt, "center": [0.05, -0.1, 0.0], "radius": 0.6}
        for tag, tr, kw in (("eval", False, {}), ("train", True, {}),
                            ("short_eval", False, {"sphere_tracing_iters": 2, "n_steps": 40, "n_secant_steps": 5}),
                            ("short_train", True, {"sphere_tracing_iters": 2, "n_steps": 40, "line_step_iters": 2})):
            for k, v in run(L, f_sph, cam, dirs, gt, tr, 3, **kw).items():
                out["%s_%s" % (tag, k)] = v
        npz("raytrace_sphere.npz", **out)

        torch.manual_seed(0)
        m_fit = O.fit_siren_to_sphere(O.SirenSDF(hidden_size=256, n_layers=3), radius=0.7, steps=300)
        f_sir = lambda x: m_fit.forward(x).sdf.reshape(-1)
        cam, dirs = pixel_rays(1500, 12, cam=(1.2, 0.4, 2.0))
        gt = silhouette(cam, dirs, (0.0, 0.0, 0.0), 0.7)
        out = {"cam": cam, "dirs": dirs, "gt": gt}
        for tag, tr, kw in (("eval", False, {}), ("train", True, {}),
                            ("short_eval", False, {"sphere_tracing_iters": 3, "n_steps": 64})):
            for k, v in run(L, f_sir, cam, dirs, gt, tr, 4, **kw).items():
                out["%s_%s" % (tag, k)] = v
        # same seed and schedule as make_golden_trace.py: the weights live in trace_siren.npz only
        import numpy as np, os
        held = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trace_siren.npz"))["siren_raw"]
        assert np.array_equal(held, siren_arrays(m_fit)["siren_raw"])
        npz("raytrace_siren.npz", **out)
    finally:
        torch.Tensor.cuda = real_cuda

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_raytrace.py
    _os.environ["ISO_GOLDEN_ONLY"] = "raytrace"
    import make_golden
    make_golden.main()
