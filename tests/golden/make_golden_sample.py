"""Golden vectors for the differentiable re-parameterisation of iso-points (Eq. 13), made by the
REFERENCE's own SampleNetwork.forward / DirectionalSamplingNetwork.forward
(DSS/models/levelset_sampling.py:1170-1207, :1370-1403) through make_golden.py's shims: the
sampled points (= the input, numerically) and the gradient of a fixed linear functional of them
w.r.t. a few of the network's parameters (the only thing these layers exist for).
usage:  ISO_GOLDEN_ONLY=sample python tests/golden/make_golden.py"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import os

import numpy as np
import torch

from oracle import iso_oracle as O
from make_golden import npz

PARAMS = ((0, "weight"), (2, "bias"), (-1, "weight"), (-1, "bias"))


def selected_grads(net, value):
    net.zero_grad()
    value.backward()
    return torch.cat([getattr(net.lins[i], n).grad.reshape(-1) for i, n in PARAMS])


def gen_sample(L):
    here = os.path.dirname(os.path.abspath(__file__))
    torch.manual_seed(0)
    net = O.fit_siren_to_sphere(O.SirenSDF(hidden_size=256, n_layers=3), radius=0.7, steps=300)
    assert np.array_equal(np.load(os.path.join(here, "trace_siren.npz"))["siren_raw"], net.raw_weights())
    g = torch.Generator().manual_seed(17)
    pts = torch.nn.functional.normalize(torch.randn(1, 600, 3, generator=g), dim=-1) * 0.7
    pts = pts + 0.01 * torch.randn(1, 600, 3, generator=g)
    cam = torch.tensor([[[0.3, 0.5, 2.4]]])
    # the points a camera sees: 1 / (D_xF . v) of the directional layer is ill-conditioned on the
    # silhouette (the reference clamps it at 1e-10), where any two f32 evaluations of D_xF disagree
    facing = (torch.nn.functional.normalize(pts, dim=-1) * torch.nn.functional.normalize(pts - cam, dim=-1)).sum(-1) < -0.3
    pts = pts[facing].view(1, -1, 3)
    w = torch.randn(1, pts.shape[1], 3, generator=g)
    ray = (pts - cam) * 1.7                                      # un-normalised on purpose (:1389)
    out, ev = L.SampleNetwork().forward(net, pts.clone(), return_eval=True)
    g_sn = selected_grads(net, (out * w).sum())
    out_d, ev_d = L.DirectionalSamplingNetwork().forward(net, pts.clone(), ray.clone(), cam.clone(), return_eval=True)
    g_dn = selected_grads(net, (out_d * w).sum())
    npz("sample_network.npz", points=pts, w=w, cam=cam, ray=ray, sn_points=out, sn_eval=ev, sn_grads=g_sn,
        dn_points=out_d, dn_eval=ev_d, dn_grads=g_dn)

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_sample.py
    _os.environ["ISO_GOLDEN_ONLY"] = "sample"
    import make_golden
    make_golden.main()
