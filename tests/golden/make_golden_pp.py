"""Third part of make_golden.py: the reference's own point_processing.upsample / wlop and
UniformProjection.insert (DSS/utils/point_processing.py:281-362, :35-122;
DSS/models/levelset_sampling.py:172-233) on small synthetic clouds.
pytorch3d's Pointclouds / knn_points and torch_cluster.fps are absent third-party code:
knn_points is shimmed by the oracle's exact brute force, wlop is run with ratio=1.0 (no FPS),
and the containers by the minimal stand-in below."""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import iso_oracle as O  # noqa: E402


class Clouds(object):
    """Stand-in for pytorch3d Pointclouds / PointClouds3D: just what wlop / insert touch."""

    def __init__(self, points_list, features_list=None):
        self.pl = [p.clone() for p in points_list]
        self.fl = features_list
        self.device = points_list[0].device

    def __len__(self):
        return len(self.pl)

    def num_points_per_cloud(self):
        return torch.tensor([len(p) for p in self.pl])

    def points_padded(self):
        mx = max(len(p) for p in self.pl)
        out = torch.zeros(len(self.pl), mx, 3)
        for i, p in enumerate(self.pl):
            out[i, : len(p)] = p
        return out

    def points_packed(self):
        return torch.cat(self.pl, 0)

    def features_packed(self):
        return torch.cat(self.fl, 0)

    def get_bounding_boxes(self):
        return torch.stack([torch.stack([p.min(0).values, p.max(0).values], dim=1) for p in self.pl])

    def clone(self):
        return Clouds(self.pl, self.fl)

    def offset_(self, off):
        s = 0
        for i, p in enumerate(self.pl):
            self.pl[i] = p + off[s:s + len(p)]
            s += len(p)
        return self

    def update_padded(self, X):
        return Clouds([X[i, : len(p)] for i, p in enumerate(self.pl)])


def gen_pp(L):
    from make_golden import npz, sphere_cloud
    import pytorch3d.ops as ops
    import pytorch3d.ops.knn as knn
    import pytorch3d.structures as st
    from collections import namedtuple
    knn._KNN = namedtuple("KNN", "dists idx knn")

    def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, return_sorted=True, **kw):
        r = O.knn_points(p1, p2, lengths1, lengths2, K=K, return_nn=True)
        return knn._KNN(r.dists, r.idx, r.knn)

    knn.knn_points = knn_points
    ops.knn_points = knn_points
    ops.is_pointclouds = lambda x: isinstance(x, Clouds)

    def list_to_padded(xs, *a, **k):
        mx = max(len(x) for x in xs)
        out = xs[0].new_zeros((len(xs), mx) + tuple(xs[0].shape[1:]))
        for i, x in enumerate(xs):
            out[i, : len(x)] = x
        return out

    st.list_to_padded = list_to_padded
    for name in ("core",):
        if "DSS." + name not in sys.modules:
            pkg = types.ModuleType("DSS." + name)
            pkg.__path__ = [os.path.join(REF, "DSS", name)]
            sys.modules["DSS." + name] = pkg
    import importlib
    PP = importlib.import_module("DSS.utils.point_processing")
    # the module was imported (with stubs bound) when levelset_sampling was loaded: rebind
    PP.knn_points = knn_points
    PP._KNN = knn._KNN
    PP.is_pointclouds = ops.is_pointclouds
    PP.list_to_padded = list_to_padded
    PP.padded_to_list = st.padded_to_list
    PP.convert_pointclouds_to_tensor = ops.convert_pointclouds_to_tensor
    L._KNN = knn._KNN

    # upsample: 700 -> 1000 points, K=16 and the K=31 used by UniformProjection.upsample
    p = sphere_cloud(700, 61)
    for K in (16, 31):
        up, n = PP.upsample(p.clone(), 1000, neighborhood_size=K)
        npz("upsample_K%d.npz" % K, points=p, n_points=1000, K=K, out_points=up, out_num=n)
    # ragged batch
    g = torch.Generator().manual_seed(5)
    pb = torch.nn.functional.normalize(torch.randn(2, 400, 3, generator=g), dim=-1)
    nb = torch.tensor([400, 400])
    upb, nnb = PP.upsample(pb.clone(), torch.tensor([460, 430]), num_points=nb, neighborhood_size=8)
    npz("upsample_batch.npz", points=pb, num_points=nb, n_points=torch.tensor([460, 430]), K=8, out_points=upb, out_num=nnb)

    # wlop with ratio 1.0 (no FPS): the perturbation is torch.randn_like under a fixed seed
    P = sphere_cloud(1500, 71, jitter=0.02)
    torch.manual_seed(123)
    out = PP.wlop(Clouds([P[0]]), ratio=1.0, neighborhood_size=16, iters=3, repulsion_mu=0.5)
    npz("wlop_ratio1.npz", points=P, seed=123, K=16, iters=3, mu=0.5, out_points=out.points_padded())

    # bilateral normal filter (FRNN neighbourhood), noisy normals on a sphere
    pn = sphere_cloud(2500, 91, jitter=0.0)
    gn = torch.Generator().manual_seed(92)
    noisy = torch.nn.functional.normalize(pn + 0.4 * torch.randn(1, 2500, 3, generator=gn), dim=-1) * 1.7
    for K, sig in ((16, 30), (30, 0.5)):
        dn = PP.denoise_normals(pn.clone(), noisy.clone(), sharpness_sigma=sig, neighborhood_size=K)
        npz("denoise_normals_K%d.npz" % K, points=pn, normals=noisy, K=K, sigma=sig, out=dn)

    # insert
    pts = sphere_cloud(3000, 81, jitter=0.0)
    g = torch.Generator().manual_seed(82)
    ref = sphere_cloud(400, 83, jitter=0.0)[0]
    met = torch.exp(3 * torch.randn(400, 1, generator=g))
    up = L.UniformProjection(knn_k=8)
    _, num_after, child, cpb = up.insert(Clouds([ref], [met]), pts.clone(), torch.tensor([3000]))
    npz("insert.npz", points=pts, ref_points=ref, ref_metrics=met, child_pts=child, child_per_batch=cpb)


def gen_idr(L):
    """The reference's own IDR-style SDF class (DSS/models/common.py:220-310, weight_norm on)
    evaluated through the reference's _compute_sdf_and_grad / _project_points."""
    import importlib
    import warnings
    from make_golden import npz
    warnings.filterwarnings("ignore")
    C = importlib.import_module("DSS.models.common")
    torch.manual_seed(3)
    H, NL, SK, NF = 128, 4, (2,), 6
    m = C.SDF(dim=3, hidden_size=H, n_layers=NL, skip_in=SK, num_frequencies=NF, weight_norm=True, bias=0.6)
    # perturb away from the geometric init so every weight matters (incl. the zeroed columns)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 1200, 3, generator=g) - 0.5) * 2
    up = L.UniformProjection()
    sdf, grad = up._compute_sdf_and_grad(x.clone(), m)
    res = up._project_points(m, x.clone(), torch.tensor([1200]), proj_max_iters=5, proj_tolerance=1e-30)
    parts = []
    for l in range(m.num_layers - 1):
        lin = getattr(m, "lin%d" % l)
        v = lin.weight_v.detach()
        W = v * (lin.weight_g.detach() / v.norm(dim=1, keepdim=True))
        parts += [W.reshape(-1), lin.bias.detach().reshape(-1)]
    npz("idr_small.npz", points=x, sdf=sdf, grad=grad, hidden=H, n_layers=NL, skip=SK[0], n_freq=NF,
        raw=torch.cat(parts), T=5, fixed_points=res.points, fixed_normals=res.normals)


def gen_siren_ref(L):
    """The reference's own Siren class (DSS/models/common.py:90-165; c_dim = 0 as test_dtu_points.py:216-227
    builds it) evaluated through the reference's _compute_sdf_and_grad / _project_points: pins SURVEY 8(a2)
    with the class the training scripts instantiate, not with the oracle's restatement of it."""
    import importlib
    import warnings
    from make_golden import npz
    warnings.filterwarnings("ignore")
    C = importlib.import_module("DSS.models.common")
    for name, H, NL, seed in (("siren_ref_128x2.npz", 128, 1, 11), ("siren_ref_256x4.npz", 256, 3, 12)):
        torch.manual_seed(seed)
        m = C.Siren(dim=3, hidden_size=H, n_layers=NL, c_dim=0, first_omega_0=30, hidden_omega_0=30.0)
        g = torch.Generator().manual_seed(seed + 100)
        x = (torch.rand(1, 1500, 3, generator=g) - 0.5) * 1.6
        up = L.UniformProjection()
        sdf, grad = up._compute_sdf_and_grad(x.clone(), m)
        res = up._project_points(m, x.clone(), torch.tensor([1500]), proj_max_iters=4, proj_tolerance=1e-30)
        parts = []
        for layer in list(m.net):
            lin = layer.linear if hasattr(layer, "linear") else layer
            parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        npz(name, points=x, sdf=sdf, grad=grad, hidden=H, n_layers=NL, raw=torch.cat(parts), T=4,
            fixed_points=res.points, fixed_normals=res.normals)

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_pp.py
    _os.environ["ISO_GOLDEN_ONLY"] = "pp,idr,siren_ref"
    import make_golden
    make_golden.main()
