"""Second half of make_golden.py: run the reference's per-point EWA set-up
(SurfaceSplatting._get_per_point_info and the methods it calls, DSS/core/rasterizer.py:344-563)
on a synthetic scene and store inputs + outputs.  Cameras / point-cloud containers are
out-of-scope pytorch3d classes; minimal stand-ins give the methods the accessors they use."""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import splat_oracle as SO  # noqa: E402  (scene generator only)


class _Transform(object):
    def __init__(self, m):
        self.m = m

    def get_matrix(self):
        return self.m


class _Cameras(object):
    def __init__(self, views, proj):
        self.views, self.proj = views, proj
        self.R = views[:, :3, :3]

    def get_full_projection_transform(self, **kw):
        return _Transform(self.views @ self.proj)


class _Clouds(object):
    def __init__(self, pts_list, nrm_list):
        self.p, self.n = pts_list, nrm_list
        self.num = torch.tensor([len(x) for x in pts_list])

    def points_packed(self):
        return torch.cat(self.p, 0)

    def normals_packed(self):
        return torch.cat(self.n, 0)

    def num_points_per_cloud(self):
        return self.num

    def cloud_to_packed_first_idx(self):
        return torch.cumsum(self.num, 0) - self.num

    def packed_to_cloud_idx(self):
        return torch.repeat_interleave(torch.arange(len(self.p)), self.num)

    def points_padded(self):
        out = torch.zeros(len(self.p), int(self.num.max()), 3)
        for i, x in enumerate(self.p):
            out[i, : len(x)] = x
        return out


def load_reference_rasterizer():
    import DSS
    for name in ("core", "utils"):
        if "DSS." + name not in sys.modules:
            pkg = types.ModuleType("DSS." + name)
            pkg.__path__ = [os.path.join(REF, "DSS", name)]
            if name == "core":          # DSS/core/__init__.py imports cameras/lighting/texture
                sys.modules["DSS.core"] = pkg
    import DSS.utils  # real helpers (gather_batch_to_packed, to_homogen ...) on stubbed imports
    import importlib
    return importlib.import_module("DSS.core.rasterizer")


def gen_splat():
    from make_golden import npz
    R = load_reference_rasterizer()
    from splat_util import sphere_scene
    S = 64
    sc = sphere_scene(3000, n_views=3, S=S, seed=21)
    num = sc["num"].tolist()
    pts_list = list(torch.split(sc["points"], num))
    nrm_list = list(torch.split(sc["normals"], num))
    clouds = _Clouds(pts_list, nrm_list)
    cams = _Cameras(sc["views"], sc["proj"])
    rs = R.PointsRasterizationSettings(image_size=S, points_per_pixel=8, cutoff_threshold=1.0,
                                       antialiasing_sigma=1.0, Vrk_isotropic=True)
    obj = object.__new__(R.SurfaceSplatting)
    obj.raster_settings, obj.cameras, obj.frnn_radius, obj._Vrk_h = rs, cams, 0.2, None
    torch.manual_seed(0)   # the tangent frame uses torch.rand_like (rasterizer.py:395-396)
    info = R.SurfaceSplatting._get_per_point_info(obj, clouds, cameras=cams, raster_settings=rs)
    npz("splat_setup.npz", points=sc["points"], normals=sc["normals"], num=sc["num"], views=sc["views"],
        proj=sc["proj"], image_size=S, cutoff=1.0, sigma=1.0, frnn_radius=0.2, Vrk_h=obj._Vrk_h.view(-1),
        radii=info["radii"], ellipse_params=info["ellipse_params"],
        cutoff_threshold=info["cutoff_threshold"], scaler=info["scaler"])
    # gather_with_neg_idx / visibility helper (utils/__init__.py:172-185, :378-399)
    from DSS.utils import gather_with_neg_idx
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(-1, 50, (2, 6, 6, 4), generator=g)
    scal = torch.rand(50, generator=g)
    out = gather_with_neg_idx(scal, 0, idx.view(-1).long().clone()).view(idx.shape)
    npz("gather_neg_idx.npz", scaler=scal, idx=idx, out=out)

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_splat.py
    _os.environ["ISO_GOLDEN_ONLY"] = "splat"
    import make_golden
    make_golden.main()
