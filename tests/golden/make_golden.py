"""Generate the golden input/output vectors under tests/golden/ by running the REFERENCE's own
Python (yifita/iso-points, mounted read-only at /root/reference) in the build container.

Only arrays are stored (inputs, seeds, expected outputs); no reference source travels.
The reference needs third-party packages that are not installed here (pytorch3d, frnn,
trimesh, ...).  They are replaced at import time by
  * inert auto-stubs for everything off the hot path, and
  * small functional shims for the handful of helpers the hot path calls
    (packed<->padded conversions; `frnn` = the oracle's brute-force contract, since
    lxxue/FRNN is an absent third-party dependency -- FRNN parity stays "unpinned").
Two statements of levelset_sampling.py that torch>=2 rejects are patched in memory
(SURVEY Appendix B): `.detach_()` on a split view (:159) and the self-indexed masked
write (:328).

usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import iso_oracle as O  # noqa: E402  (shim provider only; never the thing tested)

STUB_TOPS = {"pytorch3d", "trimesh", "skimage", "matplotlib", "plyfile", "imageio", "pymeshlab",
             "point_cloud_utils", "git", "torch_batch_svd", "easydict", "frnn", "prefix_sum",
             "tensorboard", "plotly", "cv2", "PIL", "scipy_unused"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (object,), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STUB_TOPS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_shims():
    sys.meta_path.insert(0, _StubFinder())
    import pytorch3d.ops as ops
    import pytorch3d.ops.knn as knn
    import pytorch3d.structures as st
    import pytorch3d.renderer.points.rasterize_points as rp
    import frnn
    from collections import namedtuple

    def convert_pointclouds_to_tensor(p):
        if torch.is_tensor(p):
            return p, torch.full((p.shape[0],), p.shape[1], dtype=torch.long)
        return p.points_padded(), p.num_points_per_cloud()

    def padded_to_list(x, split_size=None):
        if split_size is None:
            return list(x.unbind(0))
        return [x[i, : int(n)] for i, n in enumerate(split_size)]

    def list_to_packed(xs):
        packed = torch.cat(xs, dim=0)
        num = torch.tensor([len(x) for x in xs], dtype=torch.long)
        first = torch.cumsum(num, 0) - num
        p2l = torch.repeat_interleave(torch.arange(len(xs)), num)
        return packed, num, first, p2l

    def packed_to_padded(inputs, first_idxs, max_size):
        flat = inputs.ndim == 1
        x = inputs[:, None] if flat else inputs
        n = first_idxs.shape[0]
        out = x.new_zeros((n, max_size, x.shape[1]))
        ends = list(first_idxs[1:].tolist()) + [x.shape[0]]
        for i, (s, e) in enumerate(zip(first_idxs.tolist(), ends)):
            out[i, : e - s] = x[s:e]
        return out[..., 0] if flat else out

    def padded_to_packed(inputs, first_idxs, num_inputs):
        flat = inputs.ndim == 2
        x = inputs[..., None] if flat else inputs
        out = x.new_zeros((num_inputs, x.shape[2]))
        ends = list(first_idxs[1:].tolist()) + [num_inputs]
        for i, (s, e) in enumerate(zip(first_idxs.tolist(), ends)):
            out[s:e] = x[i, : e - s]
        return out[..., 0] if flat else out

    ops.convert_pointclouds_to_tensor = convert_pointclouds_to_tensor
    ops.packed_to_padded = packed_to_padded
    ops.padded_to_packed = padded_to_packed
    ops.eyes = lambda dim, N, device=None, dtype=torch.float32: torch.eye(dim, dtype=dtype).expand(N, dim, dim).clone()
    knn._KNN = namedtuple("KNN", "dists idx knn")
    st.padded_to_list = padded_to_list
    st.list_to_packed = list_to_packed
    rp.kMaxPointsPerBin = 22

    def frnn_grid_points(p1, p2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None, return_nn=False, **kw):
        return O.frnn_grid_points(p1, p2, lengths1, lengths2, K=K, r=r, return_nn=return_nn)

    frnn.frnn_grid_points = frnn_grid_points

    import pytorch3d.renderer.utils as ru

    def convert_to_tensors_and_broadcast(*args, dtype=torch.float32, device="cpu"):
        """pytorch3d semantics: tensors of batch size 1 or N, expanded along dim 0 to N."""
        ts = [a if torch.is_tensor(a) else torch.tensor(a, dtype=dtype, device=device) for a in args]
        n = max(t.shape[0] for t in ts)
        assert all(t.shape[0] in (1, n) for t in ts)
        return [t.expand((n,) + tuple(t.shape[1:])) if t.shape[0] != n else t for t in ts]

    ru.convert_to_tensors_and_broadcast = convert_to_tensors_and_broadcast
    frnn.frnn_gather = lambda x, idx, lengths=None: O.frnn_gather(x, idx)


def load_reference_levelset():
    """DSS.models.levelset_sampling with the two torch>=2 incompatibilities patched in memory."""
    sys.path.insert(0, REF)
    import DSS  # light __init__
    DSS._C = _Stub("DSS._C")
    sys.modules["DSS._C"] = DSS._C
    pkg = types.ModuleType("DSS.models")
    pkg.__path__ = [os.path.join(REF, "DSS", "models")]
    sys.modules["DSS.models"] = pkg  # skip DSS/models/__init__.py (imports the whole training stack)
    src = open(os.path.join(REF, "DSS", "models", "levelset_sampling.py")).read()
    a = "net_input.detach_().requires_grad_(True)"
    b = "not_converged[not_converged] = curr_not_converged"
    assert src.count(a) >= 1 and src.count(b) == 1
    src = src.replace(a, "net_input = net_input.detach().requires_grad_(True)", 1)  # :159 only
    src = src.replace(b, "not_converged[not_converged.clone()] = curr_not_converged")
    mod = types.ModuleType("DSS.models.levelset_sampling")
    mod.__package__ = "DSS.models"
    mod.__file__ = "<reference levelset_sampling.py, 2 statements patched>"
    sys.modules["DSS.models.levelset_sampling"] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    return mod


def sphere_cloud(P, seed, jitter=0.05):
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    return p + jitter * (torch.rand(1, P, 3, generator=g) - 0.5)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024))


def siren_arrays(m):
    return {"siren_raw": m.raw_weights(), "siren_hidden": m.hidden_size, "siren_layers": m.n_layers}


def gen_projection(L):
    UP = L.UniformProjection
    # cfg 1 (BASELINE.json configs[0]): 10k points in the cube, analytic unit sphere, one Newton step
    torch.manual_seed(0)
    pts = (torch.rand(1, 10000, 3) - 0.5) * 2
    num = torch.tensor([10000])
    sph = O.SphereSDF()
    r1 = UP(proj_max_iters=1)._project_points(sph, pts.clone(), num, proj_max_iters=1)
    r10 = UP()._project_points(sph, pts.clone(), num, proj_max_iters=10)
    npz("proj_sphere_cfg1.npz", points=pts, T1_points=r1.points, T1_normals=r1.normals, T1_mask=r1.mask,
        T10_points=r10.points, T10_normals=r10.normals, T10_mask=r10.mask)
    # ragged batch, offset sphere
    g = torch.Generator().manual_seed(3)
    ptsb = (torch.rand(3, 500, 3, generator=g) - 0.5) * 2
    numb = torch.tensor([500, 1, 233])
    sph2 = O.SphereSDF((0.1, -0.2, 0.05), 0.7)
    rb = UP()._project_points(sph2, ptsb.clone(), numb, proj_max_iters=6)
    npz("proj_sphere_ragged.npz", points=ptsb, num_points=numb, center=[0.1, -0.2, 0.05], radius=0.7, T=6,
        out_points=rb.points, out_normals=rb.normals, out_mask=rb.mask)
    # SIREN (small, random weights => fixed iteration count; and fitted => converging)
    torch.manual_seed(1)
    m_small = O.SirenSDF(hidden_size=64, n_layers=2)
    x = sphere_cloud(1500, 11)
    n = torch.tensor([1500])
    sdf, grad = UP()._compute_sdf_and_grad(x.clone(), m_small)
    rs = UP()._project_points(m_small, x.clone(), n, proj_max_iters=4)
    npz("proj_siren_small.npz", points=x, sdf=sdf, grad=grad, T=4, out_points=rs.points,
        out_normals=rs.normals, out_mask=rs.mask, **siren_arrays(m_small))
    torch.manual_seed(0)
    m_fit = O.fit_siren_to_sphere(O.SirenSDF(hidden_size=256, n_layers=3), steps=200)
    x = sphere_cloud(2000, 12)
    n = torch.tensor([2000])
    rf = UP()._project_points(m_fit, x.clone(), n, proj_max_iters=10)
    rf0 = UP()._project_points(m_fit, x.clone(), n, proj_max_iters=10, proj_tolerance=1e-30)
    npz("proj_siren_fitted.npz", points=x, T=10, out_points=rf.points, out_normals=rf.normals,
        out_mask=rf.mask, fixed_points=rf0.points, fixed_normals=rf0.normals, **siren_arrays(m_fit))
    return m_fit


def gen_resample(L, m_fit):
    UP = L.UniformProjection
    P = 2000
    pts = sphere_cloud(P, 41)
    num = torch.tensor([P])
    sph = O.SphereSDF()
    for tag, model, iters in (("sphere", sph, 1), ("sphere3", sph, 3), ("siren", m_fit, 1)):
        up = UP(knn_k=8)
        r0 = up._project_points(model, pts.clone(), num, proj_max_iters=10)
        rs = up.resample(model, r0.points, r0.normals, num, sample_iters=iters)
        extra = siren_arrays(m_fit) if tag == "siren" else {}
        npz("resample_%s.npz" % tag, points=pts, sample_iters=iters, knn_k=8, proj_points=r0.points,
            proj_normals=r0.normals, proj_mask=r0.mask, out_points=rs.points, out_normals=rs.normals,
            out_mask=rs.mask, knn_idx=up._knn_idx, **extra)
    # the driver with filtering (some points never converge in 3 iterations)
    g = torch.Generator().manual_seed(51)
    far = (torch.rand(1, 1500, 3, generator=g) - 0.5) * 2.6
    out = UP(proj_max_iters=3, knn_k=8).project_points(far.clone(), sph, skip_upsampling=True)
    npz("project_points_driver.npz", points=far, T=3, knn_k=8, levelset_points=out["levelset_points"],
        levelset_normals=out["levelset_normals"], mask=out["mask"])


def main():
    install_shims()
    L = load_reference_levelset()
    only = [x for x in os.environ.get("ISO_GOLDEN_ONLY", "").split(",") if x]      # parts to regenerate (all if empty)

    def want(part):
        return not only or part in only
    if want("levelset"):
        m_fit = gen_projection(L)
        gen_resample(L, m_fit)
    if want("pp"):
        from make_golden_pp import gen_pp
        gen_pp(L)
    if want("idr"):
        from make_golden_pp import gen_idr
        gen_idr(L)
    if want("siren_ref"):
        from make_golden_pp import gen_siren_ref
        gen_siren_ref(L)
    if want("trace"):
        from make_golden_trace import gen_trace
        gen_trace(L)
    if want("raytrace"):
        from make_golden_raytrace import gen_raytrace
        gen_raytrace(L)
    if want("image"):
        from make_golden_image import gen_image
        gen_image(L)
    if want("sample"):
        from make_golden_sample import gen_sample
        gen_sample(L)
    if want("ear"):
        from make_golden_ear import gen_ear
        gen_ear(L)
    if want("splat"):
        from make_golden_splat import gen_splat
        gen_splat()


if __name__ == "__main__":
    main()
