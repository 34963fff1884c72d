"""Golden vectors for the edge-aware resampler, made by the REFERENCE's own
EdgeAwareProjection._create_tree / .denoise_normals / .upsample
(DSS/models/levelset_sampling.py:442-661) through make_golden.py's shims, with the oracle's
brute-force K-nearest search standing in for pytorch3d.knn_points (absent third-party code).
usage:  ISO_GOLDEN_ONLY=ear python tests/golden/make_golden.py"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

from collections import namedtuple

import torch

from oracle import iso_oracle as O
from make_golden import npz


def bind(L):
    KNN = namedtuple("KNN", "dists idx knn")

    def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, return_sorted=True, **kw):
        r = O.knn_points(p1, p2, lengths1, lengths2, K=K, return_nn=True)
        return KNN(r.dists, r.idx, r.knn)

    def list_to_padded(xs, *a, **k):
        mx = max(len(x) for x in xs)
        out = xs[0].new_zeros((len(xs), mx) + tuple(xs[0].shape[1:]))
        for i, x in enumerate(xs):
            out[i, : len(x)] = x
        return out

    L.knn_points, L._KNN = knn_points, KNN
    L.knn_gather = lambda x, idx, lengths=None: O.knn_gather(x, idx, lengths)
    L.list_to_padded = list_to_padded
    L.padded_to_list = lambda x, split: [x[i, :n] for i, n in enumerate(split)]


def gen_ear(L):
    bind(L)
    box = O.BoxSDF()
    g = torch.Generator().manual_seed(41)
    p = (torch.rand(1, 900, 3, generator=g) - 0.5) * 1.3
    num = torch.tensor([900])
    pts = O.project_points(box, p, num, proj_max_iters=10).points
    for tag, kw in (("K16", dict(knn_k=16, upsample_ratio=1.2)), ("K31_sharp", dict(knn_k=31, sharpness_angle=30, edge_sensitivity=2,
                                                               upsample_ratio=1.3, repulsion_mu=0.3))):
        ear = L.EdgeAwareProjection(**kw)
        ear._create_tree(pts.clone(), refresh_tree=True, num_points_per_cloud=num)
        _, n0 = ear._compute_sdf_and_grad(pts.clone(), box)
        nd, wp, wn = ear.denoise_normals(pts.clone(), n0.clone(), num)
        up, n_up = ear.upsample(pts.clone(), 900, box, num.clone())
        npz("ear_%s.npz" % tag, points=pts, normals=n0, denoised=nd, weights_p=wp, weights_n=wn, out_points=up, out_num=n_up,
            **{"kw_" + k: v for k, v in kw.items()})
    # the inherited driver (project -> resample on the K-nearest tree -> edge-aware upsample -> project)
    sph = O.SphereSDF()
    far = torch.nn.functional.normalize(torch.randn(1, 1200, 3, generator=g), dim=-1) * (1 + 0.1 * (torch.rand(1, 1200, 1, generator=g) - 0.5))
    kw = dict(knn_k=12, sample_iters=2, upsample_ratio=1.1)
    out = L.EdgeAwareProjection(**kw).project_points(far.clone(), sph)
    npz("ear_driver.npz", points=far, levelset_points=out["levelset_points"], levelset_normals=out["levelset_normals"],
        mask=out["mask"], **{"kw_" + k: v for k, v in kw.items()})

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_ear.py
    _os.environ["ISO_GOLDEN_ONLY"] = "ear"
    import make_golden
    make_golden.main()
