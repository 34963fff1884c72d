"""Host-side mirror of the reference's iso-point operators
(DSS/models/levelset_sampling.py): same class, method names, arguments, return
types and error behaviour; the arithmetic runs in libisopoints_hip.so.

  UniformProjection._project_points   levelset_sampling.py:290-351
  UniformProjection.resample          levelset_sampling.py:239-288
  UniformProjection._create_tree      levelset_sampling.py:110-140
  UniformProjection.project_points    levelset_sampling.py:353-439
  SphereTracing.project_points        levelset_sampling.py:663-808
  find_zero_crossing_between_point_pairs / run_Secant_method   levelset_sampling.py:1210-1367

Model dispatch: an analytic sphere and SIREN networks run in the fused
HIP kernels; any other nn.Module takes the generic route -- the reference's own
algorithm (model.forward + autograd.grad on compacted active points) on the GPU.
"""
import math
from collections import namedtuple
from typing import Optional

import os as _os
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import frnn
from .sdf_models import FusedSdf, PackedIdr, PackedSiren, idr_spec, siren_spec

ProjectionResult = namedtuple("ProjectionResult", ("points", "normals", "mask"))


# ----------------------------------------------------------------------------- helpers
def eps_denom(denom, eps=1e-17):
    """DSS/utils/mathHelper.py:14-18."""
    denom_sign = denom.sign() + (denom == 0.0).type_as(denom)
    return denom_sign * torch.clamp(denom.abs(), eps)


def with_host_lengths(num_points, host):
    """Attach the host copy of a lengths tensor so later stages need no .tolist() sync (dropped as
    soon as the tensor is modified in place: the copy is keyed on the tensor's version)."""
    num_points._iso_host = [int(x) for x in host]
    num_points._iso_host_version = num_points._version
    return num_points


def host_lengths(num_points):
    h = getattr(num_points, "_iso_host", None)
    if h is None or getattr(num_points, "_iso_host_version", None) != num_points._version:
        h = [int(x) for x in num_points.tolist()]  # host sync, as the reference (:308)
        with_host_lengths(num_points, h)
    return h


def true_counts(mask):
    """Host list of the number of True entries per cloud of a (B, P) bool mask: one device reduction + one host read,
    remembered on the tensor (keyed on its version, like with_host_lengths)."""
    h = getattr(mask, "_iso_true", None)
    if h is None or getattr(mask, "_iso_true_version", None) != mask._version:
        B = mask.shape[0]
        h = [int(x) for x in torch.count_nonzero(mask.reshape(B, -1), dim=1).tolist()]
        mask._iso_true, mask._iso_true_version = h, mask._version
    return h


def full_lengths(points):
    B, P = points.shape[0], points.shape[1]
    return with_host_lengths(torch.full((B,), P, dtype=torch.long, device=points.device), [P] * B)


def cloud_diag(points_padded, lengths=None):
    """|bbox diagonal| of each cloud, (N,) device tensor (iso_points_bbox; no host sync).
    lengths=None spans the whole padded tensor, like the reference's max/min over dim 1."""
    N, P = points_padded.shape[0], points_padded.shape[1]
    pts = points_padded.detach().float().contiguous()
    mm = torch.empty((N, 8), dtype=torch.float32, device=pts.device)
    _lib.call("iso_points_bbox", _lib.ptr(pts), _lib.ptr(lengths) if lengths is not None else None, N, P,
              _lib.ptr(mm), _lib.stream())
    return (mm[:, 4:7] - mm[:, 0:3]).norm(dim=-1)


def convert_pointclouds_to_tensor(pcl):
    """pytorch3d.ops.utils.convert_pointclouds_to_tensor for the two inputs the reference
    passes (levelset_sampling.py:374): a padded tensor or a Pointclouds-like object."""
    if torch.is_tensor(pcl):
        return pcl, full_lengths(pcl)
    if hasattr(pcl, "points_padded") and hasattr(pcl, "num_points_per_cloud"):
        return pcl.points_padded(), pcl.num_points_per_cloud()
    raise ValueError("The inputs should be either a Pointclouds object or a torch.Tensor")


def padded_to_packed(padded, lens):
    B, P = padded.shape[0], padded.shape[1]
    if B == 1:
        return padded[0, : lens[0]]
    if all(l == P for l in lens):
        return padded.reshape((B * P,) + tuple(padded.shape[2:]))
    return torch.cat([padded[b, : lens[b]] for b in range(B)], dim=0)


def packed_to_padded(packed, lens, pad_value=0):
    B = len(lens)
    mx = max(lens) if B else 0
    if B == 1:
        return packed.view((1, mx) + tuple(packed.shape[1:]))
    if all(l == mx for l in lens):
        return packed.view((B, mx) + tuple(packed.shape[1:]))
    out = packed.new_full((B, mx) + tuple(packed.shape[1:]), pad_value)
    s = 0
    for b, l in enumerate(lens):
        out[b, :l] = packed[s : s + l]
        s += l
    return out


def reduce_mask_padded(values, mask):
    """DSS/utils/__init__.py:149-169: drop the masked-out rows of each cloud, re-pad with 0."""
    counts = true_counts(mask)                       # one host read per mask tensor, not one per call
    packed = values[mask]
    return packed_to_padded(packed, counts)


def _filter_projection_result(result):
    """levelset_sampling.py:59-65."""
    points, normals, mask = result
    return ProjectionResult(reduce_mask_padded(points, mask), reduce_mask_padded(normals, mask),
                            reduce_mask_padded(mask, mask))


# ----------------------------------------------------------------------------- the operator
class LevelSetProjection(object):
    def __init__(self, proj_max_iters=10, proj_tolerance=5.0e-5, max_points_per_pass=120000):
        self.proj_max_iters = proj_max_iters
        self.proj_tolerance = proj_tolerance
        self.max_points_per_pass = max_points_per_pass

    def project_points(self, points_init, network, latent, levelset):
        raise NotImplementedError


_GENERIC_WARNED = set()


def _warn_generic_route(model, has_kwargs):
    """Once per model class: this network has no fused SDF kernel (SIREN 3 -> H <= 256 x <= 8 sine layers -> 1 and
    IDR-style 128 / 256 / 512 do), so the projection runs the reference's torch loop (model.forward + autograd per
    iteration).  Loud on purpose: it is 10-100x slower and it is not the path the benchmarks measure."""
    key = (type(model).__name__, has_kwargs)
    if key in _GENERIC_WARNED:
        return
    _GENERIC_WARNED.add(key)
    import warnings
    warnings.warn("iso_points_amd: no fused SDF kernel for %s%s -- UniformProjection falls back to the generic torch "
                  "autograd loop (levelset_sampling.py:306-341), far slower than the fused path"
                  % (type(model).__name__, " called with forward kwargs" if has_kwargs else ""), RuntimeWarning, stacklevel=3)


class UniformProjection(LevelSetProjection):
    """Same constructor as the reference (levelset_sampling.py:92-108)."""

    def __init__(self, proj_max_iters=10, proj_tolerance=5e-5, max_points_per_pass=120000,
                 sample_iters=1, knn_k=8, resampling_clip=0.02, **kwargs):
        super().__init__(proj_max_iters=proj_max_iters, proj_tolerance=proj_tolerance,
                         max_points_per_pass=max_points_per_pass)
        self.knn_k = knn_k
        self.sample_iters = sample_iters
        self.resampling_clip = resampling_clip  # stored, unused -- as in the reference (:108)
        self._packed_cache = None
        self._pack_scope, self._pack_depth = 0, 0      # see _packing(): one weight image per operator call
        self._packed_for = None
        self.reuse_packed = False      # keep the packed weight image of the last call (the caller clears
                                       # _packed_cache whenever the weights may have changed)
        self.materialize_knn = True    # resample(): also write the neighbour lists the fused kernel selects
                                       # (_knn_idx / _knn_dists, as the reference's _create_tree leaves them)

    # -- tree ------------------------------------------------------------------------
    def _create_tree(self, points_padded, refresh_tree=True, num_points_per_cloud=None):
        """Neighbour lists of the cloud within r = knn_k * sqrt(bbox diagonal / P) (levelset_sampling.py:110-140):
        a self-inclusive K = knn_k + 1 query whose first column (the point itself) is dropped.  Cached on self
        under the reference's attribute names."""
        if refresh_tree or getattr(self, "_knn_idx", None) is None:
            assert points_padded.ndim == 3
            lengths = full_lengths(points_padded) if num_points_per_cloud is None else num_points_per_cloud
            radius = self.knn_k * torch.sqrt(cloud_diag(points_padded) / lengths.float())
            d2, ids, xyz, _ = frnn.frnn_grid_points(points_padded, points_padded, lengths, lengths, K=self.knn_k + 1,
                                                    r=radius, grid=None, return_nn=True)
            self._knn_idx, self._knn_dists, self._knn_nn = ids[..., 1:], d2[..., 1:], xyz[..., 1:, :]
            self._knn_gather = self.knn_gather = frnn.frnn_gather
        return self._knn_idx

    # -- SDF evaluation ----------------------------------------------------------------
    def _compute_sdf_and_grad(self, points, model, latent=None, **forward_kwargs):
        """levelset_sampling.py:142-170 (generic route: chunks of max_points_per_pass)."""
        shp = points.shape
        points_packed = points.reshape(-1, 3)
        if siren_spec(model) is not None and points_packed.is_cuda and not forward_kwargs:
            from .sdf_models import siren_sdf_and_grad
            sdf, grad = siren_sdf_and_grad(model, points_packed)
            return sdf.view(shp[:-1]), grad.view(shp)
        if idr_spec(model) is not None and points_packed.is_cuda and not forward_kwargs:
            from .sdf_models import idr_sdf_and_grad
            sdf, grad = idr_sdf_and_grad(model, points_packed)
            return sdf.view(shp[:-1]), grad.view(shp)
        grads, evals = [], []
        with torch.no_grad():
            model.eval()
            for sub in torch.split(points_packed, self.max_points_per_pass, dim=0):
                with torch.enable_grad():
                    x = sub.detach().requires_grad_(True)
                    out = model.forward(x, **forward_kwargs).sdf
                    (g,) = torch.autograd.grad([out], [x], torch.ones_like(out), retain_graph=False)
                grads.append(g)
                evals.append(out.detach())
            if not grads:
                return points_packed.new_zeros(shp[:-1]), points_packed.new_zeros(shp)
            return torch.cat(evals, 0).view(shp[:-1]), torch.cat(grads, 0).view(shp)

    # -- projection --------------------------------------------------------------------
    def _packing(self):
        """with self._packing(): the projections issued inside share one packed weight image (nested uses join the outer
        scope)."""
        owner = self

        class _Scope(object):
            def __enter__(self_):
                if owner._pack_depth == 0:
                    owner._pack_scope += 1
                owner._pack_depth += 1

            def __exit__(self_, *exc):
                owner._pack_depth -= 1
                return False
        return _Scope()

    def _project_packed(self, model, pts, proj_max_iters, proj_tolerance, follow=None, **forward_kwargs):
        """pts (n,3) f32 contiguous on the GPU -> points, normals, mask(bool).  follow: bricks.Follow -- side work of
        the launch for the next stages of the cycle (iso_follow); returns False in `follow.done` when this model's route
        cannot do it (the caller then runs the stand-alone passes)."""
        n = pts.shape[0]
        dev = pts.device
        out = torch.empty_like(pts)
        # every point is evaluated at least once and the fused kernels write normals and mask for each point they
        # evaluate (csrc/iso_newton.h: iso_step_finish; k_project_sphere): no zero fill; the mask is written as 0 / 1
        # bytes straight into a bool tensor (no uint8 -> bool conversion pass)
        normals = torch.empty_like(pts)
        mask = torch.empty((n,), dtype=torch.bool, device=dev)
        p = _lib.ptr
        if getattr(model, "iso_analytic", None) == "sphere" and not forward_kwargs:
            cached = getattr(model, "_iso_center", None)     # host copy of the centre, re-read when the buffer changes
            if cached is None or cached[0] != (model.center._version, model.center.data_ptr()):
                cached = ((model.center._version, model.center.data_ptr()), [float(x) for x in model.center.tolist()])
                model._iso_center = cached
            c = cached[1]
            if follow is not None:
                import ctypes
                fs = follow.struct()
                _lib.call("iso_project_sphere_follow", p(pts), p(out), p(normals), p(mask), n, c[0], c[1], c[2],
                          float(model.radius), int(proj_max_iters), float(proj_tolerance), ctypes.byref(fs), _lib.stream())
                follow.done = True
                return out, normals, mask
            _lib.call("iso_project_sphere", p(pts), p(out), p(normals), p(mask), n, c[0], c[1], c[2],
                      float(model.radius), int(proj_max_iters), float(proj_tolerance), _lib.stream())
            return out, normals, mask
        if follow is not None:
            follow.done = False
        if siren_spec(model) is not None and not forward_kwargs:
            # the packed weight image is re-used on the caller's word (reuse_packed), or INSIDE one call of an operator
            # of this class (project_points / resample run several projections on weights that cannot change in between:
            # _pack_scope); never across calls on the strength of (storage, version) alone -- in-place updates through
            # `.data` (EMA copies, weight clipping) and raw-pointer writes bump neither
            pc = self._packed_cache
            in_scope = self._pack_depth > 0 and getattr(pc, "_scope", None) == self._pack_scope
            ps = pc if (isinstance(pc, PackedSiren) and self._packed_for is model
                        and (self.reuse_packed or (in_scope and pc.current(model, dev)))) else PackedSiren(model, dev)
            ps._scope = self._pack_scope if self._pack_depth > 0 else None
            self._packed_for = model
            ws = ps.workspace(n)
            _lib.call("iso_project_siren", p(pts), p(out), p(normals), p(mask), n, p(ps.packed),
                      ps.hidden, ps.n_hidden, ps.omega_first, ps.omega_hidden, int(proj_max_iters),
                      float(proj_tolerance), p(ws), ws.numel(), _lib.stream())
            self._packed_cache = ps  # keep the workspace alive until the stream has used it
            return out, normals, mask
        if idr_spec(model) is not None and not forward_kwargs:
            pk = self._packed_cache if (self.reuse_packed and isinstance(self._packed_cache, PackedIdr)
                                        and self._packed_for is model) else PackedIdr(model, dev)
            self._packed_for = model
            ws = pk.workspace(n)
            _lib.call("iso_project_idr", p(pts), p(out), p(normals), p(mask), n, p(pk.packed), pk.hidden,
                      pk.n_layers, pk.skip, pk.n_freq, 100.0, int(proj_max_iters), float(proj_tolerance),
                      p(ws), ws.numel(), _lib.stream())
            self._packed_cache = pk
            return out, normals, mask
        _warn_generic_route(model, bool(forward_kwargs))
        return self._project_packed_generic(model, pts, proj_max_iters, proj_tolerance, **forward_kwargs)

    def _project_packed_generic(self, model, pts, proj_max_iters, proj_tolerance, **forward_kwargs):
        """The reference's loop (levelset_sampling.py:309-344) for models without a fused kernel."""
        points_packed = pts.clone()
        not_converged = torch.ones(points_packed.shape[0], dtype=torch.bool, device=pts.device)
        normals_packed = torch.zeros_like(points_packed)
        it = 0
        while True:
            curr_points = points_packed[not_converged]
            curr_sdf, curr_grad = self._compute_sdf_and_grad(curr_points, model, **forward_kwargs)
            normals_packed[not_converged] = curr_grad
            curr_not_converged = curr_sdf.reshape(-1).abs() > proj_tolerance
            nc = not_converged.clone()
            nc[not_converged] = curr_not_converged
            not_converged = nc
            if (~not_converged).all() or it == proj_max_iters:
                break
            it += 1
            active_grad = curr_grad[curr_not_converged]
            active_sdf = curr_sdf.reshape(-1)[curr_not_converged]
            active_pts = curr_points[curr_not_converged]
            ssg = torch.sum(active_grad ** 2, dim=-1, keepdim=True)
            move = active_sdf.view(-1, 1) * (active_grad / eps_denom(ssg, 1.0e-17))
            move = F.normalize(move, dim=-1, eps=1e-15) * move.norm(dim=-1, keepdim=True).clamp_max(0.1)
            points_packed[not_converged] = active_pts - move
        return points_packed, normals_packed, ~not_converged

    def _project_points(self, model, points, num_points, proj_max_iters=None, proj_tolerance=None, follow=None,
                        **forward_kwargs) -> ProjectionResult:
        """points (B,P,3) padded, num_points (B,) -> ProjectionResult(points, normals, mask).  follow (not in the
        reference's signature; one full cloud only): see _project_packed."""
        proj_max_iters = proj_max_iters or self.proj_max_iters
        proj_tolerance = proj_tolerance or self.proj_tolerance
        if not points.is_cuda:
            raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
        lens = host_lengths(num_points)
        packed = padded_to_packed(points.detach().float(), lens).contiguous()
        with torch.no_grad():
            pts, normals, valid = self._project_packed(model, packed, proj_max_iters, proj_tolerance, follow=follow,
                                                       **forward_kwargs)
        return ProjectionResult(packed_to_padded(pts, lens), packed_to_padded(normals, lens),
                                packed_to_padded(valid, lens, pad_value=False))

    # -- insert / upsample ---------------------------------------------------------------
    def insert(self, ref_pcl, points, num_points, current_knn_result=None):
        """Children around the iso-points nearest to the high-metric reference points
        (levelset_sampling.py:172-233), on the device: iso_insert_fathers / iso_insert_children
        (csrc/insert.hip).  ref_pcl: one cloud with points_packed() (R,3) and features_packed() (R,1), the
        per-point metric.  Returns (points + children, num_points + children, children padded, children
        per cloud).  One host read, of the result sizes."""
        B = points.shape[0]
        try:
            return self._insert(ref_pcl, points, num_points, current_knn_result)
        except Exception as e:                                        # :226-229: any failure -> no children, logged
            import logging
            logging.getLogger(__name__).error("Error occurred during insertion %s", e)
            return points, num_points, points.new_zeros((B, 0, 3)), with_host_lengths(num_points.new_zeros((B,)), [0] * B)

    def _insert(self, ref_pcl, points, num_points, current_knn_result=None):
        from . import bricks
        PATCH = 8                                                     # neighbours that mother a child (:181)
        B, P = points.shape[0], points.shape[1]
        dev = points.device
        pts = points.detach().float().contiguous()
        metric = ref_pcl.features_packed().detach().float().reshape(-1)
        ref_xyz = ref_pcl.points_packed().detach().float()
        R = int(ref_xyz.shape[0])
        no_children = (points.new_zeros((B, 0, 3)), with_host_lengths(num_points.new_zeros((B,)), [0] * B))
        if R == 0 or P == 0:
            return points, num_points, no_children[0], no_children[1]
        # device-side scalars: spacing of the cloud, query radius (:178-183)
        box = bricks.points_bbox(pts.view(-1, 3))
        spacing = torch.sqrt((box[4:7] - box[0:3]).norm() / R)
        radius = torch.clamp(spacing * PATCH, max=0.2)
        if current_knn_result is None:
            _, nbr, _, _ = frnn.frnn_grid_points(pts, pts, num_points, num_points, K=PATCH + 1,
                                                 r=radius.expand(B).contiguous(), grid=None, return_nn=False)
            nbr = nbr[..., 1:]
        else:
            nbr = current_knn_result.idx
        nbr = nbr.to(torch.int64).contiguous()
        # the selected reference points (:186-194): those above min(2 median, max / 2) when they are between 1 and
        # `cap` many, else the `cap` largest -- either way the n_sel largest metrics, n_sel known on the device only
        cap = min(50, R // 20)
        kmax = max(cap, 1)
        bar = torch.minimum(metric.median() * 2, metric.max() * 0.5)
        above = (metric > bar).sum()
        n_sel = torch.where((above >= 1) & (above <= cap), above, torch.full_like(above, kmax)).to(torch.int32).view(1)
        # the kmax largest, ties at the cut resolved like the reference's ascending sort()[-kmax:] (:193-194)
        selected = ref_xyz[torch.sort(metric, stable=True).indices[-kmax:].flip(0)].contiguous()
        limits = torch.stack([(radius * 4) ** 2, 4 * spacing * spacing]).float().contiguous()
        father = torch.empty((B, P), dtype=torch.uint8, device=dev)
        lens_dev = num_points.to(torch.int64).contiguous()
        cp = _lib.ptr
        _lib.call("iso_insert_fathers", cp(pts), cp(lens_dev), B, P, cp(selected), cp(n_sel), cp(limits), cp(father),
                  _lib.stream())
        rank = father.to(torch.int64).cumsum(dim=1).contiguous()
        per_cloud = rank[:, -1] * PATCH
        first = (per_cloud.cumsum(0) - per_cloud).contiguous()
        sizes = [int(v) for v in per_cloud.tolist()]                  # the result shapes: the one host read
        total = sum(sizes)
        if total == 0:
            return points, num_points, no_children[0], no_children[1]
        packed = torch.empty((total, 3), dtype=torch.float32, device=dev)
        _lib.call("iso_insert_children", cp(pts), cp(nbr), B, P, nbr.shape[-1], PATCH, cp(father), cp(rank), cp(first),
                  cp(packed), _lib.stream())
        children = packed_to_padded(packed, sizes)
        grown = torch.cat((points, children.to(points.dtype)), dim=1)
        per_cloud = with_host_lengths(per_cloud, sizes)
        return grown, num_points + per_cloud, children, per_cloud

    def upsample(self, points, n_points, model, num_points=None, **forward_kwargs):
        """levelset_sampling.py:235-237."""
        from .point_processing import upsample as _upsample
        return _upsample(points, n_points, num_points=num_points, neighborhood_size=31)

    # -- resample ----------------------------------------------------------------------
    def repulsion_step(self, points, normals_init, idx, inv_sigma, first_point=0):
        """One tangent-plane repulsion move (levelset_sampling.py:268-284) of a single cloud.
        points (1,P,3); normals_init un-normalised (normalised on the fly in the kernel);
        idx (1,n,K) int64 view (row stride may exceed K) for the n points starting at
        `first_point` (n = P, first_point = 0 unless the cloud is sharded over ranks);
        inv_sigma: device scalar.  Returns the moved (1,n,3) points."""
        assert points.shape[0] == 1, "resample supports one cloud (as the reference, :256,:274)"
        n, K = idx.shape[1], idx.shape[-1]
        pts = points[0].contiguous()
        nrm = normals_init[0].contiguous()
        assert idx.stride(-1) == 1 and first_point + n <= pts.shape[0]
        out = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
        _lib.call("iso_repulse", _lib.ptr(pts), _lib.ptr(nrm), _c_ptr(idx), int(idx.stride(-2)),
                  _lib.ptr(out), n, int(first_point), K, _lib.ptr(inv_sigma), _lib.stream())
        return out.view(1, n, 3)

    def resample(self, model, points_init, normals_init, num_points, sample_iters=None,
                 **forward_kwargs) -> ProjectionResult:
        sample_iters = sample_iters or self.sample_iters
        batch_size = points_init.shape[0]
        if num_points is None:
            num_points = full_lengths(points_init)
        if sample_iters == 0 or points_init.nelement() < 2 * (self.knn_k + 1):
            return ProjectionResult(points_init, normals_init,
                                    points_init.new_full(points_init.shape[:-1], True, dtype=torch.bool))
        if batch_size != 1:
            raise NotImplementedError("resample: one cloud per call (the reference's broadcast of "
                                      "inv_sigma_spatial is only valid for batch size 1, :256,:274)")
        points = points_init
        projection_result = None
        idx = None
        inv_sigma = None
        P = points_init.shape[1]
        fused = (points_init.is_cuda and self.knn_k + 1 <= 13 and host_lengths(num_points)[0] == P
                 and points_init.dtype == torch.float32)
        scope = self._packing()
        scope.__enter__()                    # (the projections of all sample iterations share one packed weight image)
        try:
            return self._resample_iterations(model, points_init, normals_init, num_points, sample_iters, fused,
                                             **forward_kwargs)
        finally:
            scope.__exit__(None, None, None)

    def _resample_iterations(self, model, points_init, normals_init, num_points, sample_iters, fused, **forward_kwargs):
        points, projection_result, idx, inv_sigma, P = points_init, None, None, None, points_init.shape[1]
        for sample_iter in range(sample_iters):
            if sample_iter == 0 and fused:
                # tree + repulsion in one kernel on the brick grid (csrc/bricks.hip); the neighbour lists are
                # only materialised when a later iteration re-uses them (:262-266) or the caller asks for them
                from . import bricks
                grid = getattr(self, "_grid", None)          # the workspace (a few hundred MB at 1 M points, cleared once)
                if grid is None or grid.n_own != P or grid.ws.device != points.device:     # is kept between calls
                    grid = self._grid = bricks.BrickGrid(P, points.device)
                grid.build(points[0].contiguous(), normals_init[0].contiguous(), knn_k=self.knn_k)
                keep = sample_iters > 1 or self.materialize_knn
                moved, idx_f, d2_f = bricks.resample_fused(grid, self.knn_k + 1, want_idx=keep)
                inv_sigma = grid.ws[48:52].view(torch.float32)                      # P / diag of points_init (:254-256)
                idx = idx_f.view(1, P, self.knn_k) if keep else None
                self._knn_idx = idx
                self._knn_dists = d2_f.view(1, P, self.knn_k) if keep else None
                self._knn_nn = None
                self.knn_gather = self._knn_gather = frnn.frnn_gather
                points = moved.view(1, P, 3)
            else:
                if inv_sigma is None:
                    diag = cloud_diag(points_init.reshape(1, -1, 3))[0]               # :254-255
                    inv_sigma = (num_points.float() / diag).reshape(1).contiguous()  # device scalar (:256)
                if sample_iter % 2 == 0:
                    self._create_tree(points, refresh_tree=True, num_points_per_cloud=num_points)
                    idx = self._knn_idx
                points = self.repulsion_step(points, normals_init, idx, inv_sigma)
            projection_result = self._project_points(model, points, num_points, proj_max_iters=3,
                                                     **forward_kwargs)
        return projection_result

    # -- driver ------------------------------------------------------------------------
    def project_points(self, point_clouds, model, normals_init: Optional[torch.Tensor] = None,
                       skip_resampling: bool = False, skip_upsampling: bool = False, ref_pcl=None,
                       proj_max_iters: Optional[int] = None, sample_iters: Optional[int] = None,
                       **forward_kwargs):
        """The whole extraction (levelset_sampling.py:353-439) as a chain of stages over one ProjectionResult:
        project -> [keep converged, resample] -> [keep converged, grow (insert around ref_pcl, or upsample back
        to the input count), project what is new].  Returns the reference's dict."""
        cloud, lengths = convert_pointclouds_to_tensor(point_clouds)
        wanted = lengths                                              # upsample() refills to the input count

        def converged(res):
            # keep the converged points (:59-65).  ONE host read -- the number of them per cloud -- decides: all of
            # them (the usual case on a fitted network) -> the result as it is, no copy; else one compaction
            n_true = true_counts(res.mask)
            if n_true == host_lengths(lengths):                       # nothing dropped: the lengths tensor as it is
                return res, lengths
            kept = _filter_projection_result(res)
            return kept, with_host_lengths(torch.tensor(n_true, dtype=torch.long, device=res.mask.device), n_true)

        with torch.no_grad(), self._packing():
            res = self._project_points(model, cloud, lengths, proj_max_iters=proj_max_iters or self.proj_max_iters,
                                       **forward_kwargs)
            optimistic = None
            if (not skip_resampling and res.mask.is_cuda and res.mask.shape[0] == 1 and not _os.environ.get("ISO_OPAPI_SYNC")
                    and host_lengths(lengths) == [int(res.mask.shape[1])] and getattr(self, "_all_converged_last", False)):
                # On a fitted network every point converges: the resampling is ISSUED on the whole result before the host
                # knows the number of converged points (the count's kernel is in the queue, its read comes after the
                # resampling has been queued, so the GPU does not wait for the host between the two stages).  A count that
                # says otherwise discards this work and takes the reference's order below.
                c1 = torch.count_nonzero(res.mask.reshape(1, -1), dim=1)
                optimistic = self.resample(model, res.points, res.normals, lengths,
                                           sample_iters=sample_iters or self.sample_iters, **forward_kwargs)
                c2 = torch.count_nonzero(optimistic.mask.reshape(1, -1), dim=1)
                both = torch.stack([c1, c2]).tolist()                   # ONE host read for both masks
                res.mask._iso_true, res.mask._iso_true_version = [int(both[0][0])], res.mask._version
                optimistic.mask._iso_true, optimistic.mask._iso_true_version = [int(both[1][0])], optimistic.mask._version
            # (the optimistic order only after a call on this object that saw every point converge: on a partly fitted
            # network it would resample -- and rebuild the grid on -- unconverged points every call for nothing: ADVICE r5)
            self._all_converged_last = true_counts(res.mask) == host_lengths(lengths)
            if sum(true_counts(res.mask)) == 0:                       # `not mask.any()` (:396), the count is re-used below
                return {"levelset_points": res.points, "mask": res.mask}
            if not skip_resampling:
                if optimistic is not None and true_counts(res.mask) == host_lengths(lengths):
                    res = optimistic
                else:
                    res, lengths = converged(res)
                    res = self.resample(model, res.points, res.normals, lengths,
                                        sample_iters=sample_iters or self.sample_iters, **forward_kwargs)
            if not skip_upsampling:
                res, lengths = converged(res)
                if ref_pcl is not None:                               # :411-424: children of the flagged regions
                    _, _, fresh, n_fresh = self.insert(ref_pcl, res.points, lengths)
                    if fresh.shape[1] > 0:
                        more = self._project_points(model, fresh, n_fresh, proj_max_iters=10, **forward_kwargs)
                        res = ProjectionResult(*(torch.cat(pair, dim=1) for pair in zip(res, more)))
                else:                                                 # :426-434: back to the input count
                    dense, lengths = self.upsample(res.points, wanted, model, lengths, **forward_kwargs)
                    res = self._project_points(model, dense, lengths, proj_max_iters=10, **forward_kwargs)
            return {"levelset_points": res.points, "levelset_normals": res.normals, "mask": res.mask}


class EdgeAwareProjection(UniformProjection):
    """Sample and project away from the edges (levelset_sampling.py:442-661): UniformProjection
    with an exact K-nearest tree, a bilateral normal filter and an upsampling step that inserts
    points where the sampling is sparse ACROSS normal discontinuities.  Same constructor, same
    `_create_tree` / `denoise_normals` / `upsample` signatures and results.  The neighbour search
    (iso_frnn_* with an unbounded radius), the gathers and D_xF (fused SDF kernels) are the
    library's; the O(P K^2) sparsity statement is one kernel (iso_ear_candidates).
    One cloud per call: the reference's `num_points / 2.0` broadcasts (:515,:551) only hold for
    batch size 1."""

    def __init__(self, proj_max_iters=10, proj_tolerance=5e-5, max_points_per_pass=120000, knn_k=31,
                 repulsion_mu=0.5, sample_iters=5, total_iters=1, sharpness_angle=15, edge_sensitivity=1,
                 resampling_clip=0.02, upsample_ratio=1.5, **kwargs):
        super().__init__(sample_iters=sample_iters, total_iters=total_iters, resampling_clip=resampling_clip,
                         knn_k=knn_k, proj_max_iters=proj_max_iters, proj_tolerance=proj_tolerance,
                         max_points_per_pass=max_points_per_pass)
        self.sharpness_sigma = 1 - math.cos(sharpness_angle / 180 * math.pi)
        self.repulsion_mu = repulsion_mu
        self.edge_sensitivity = edge_sensitivity
        self.upsample_ratio = upsample_ratio

    def _create_tree(self, points_padded, refresh_tree=True, num_points_per_cloud=None):
        """:471-498 -- K+1 nearest neighbours of every point among the cloud, self dropped."""
        from .point_processing import knn_points
        if not refresh_tree and getattr(self, "_knn_idx", None) is not None:
            return self._knn_idx
        assert points_padded.ndim == 3
        if num_points_per_cloud is None:
            num_points_per_cloud = full_lengths(points_padded)
        res = knn_points(points_padded, points_padded, num_points_per_cloud, num_points_per_cloud,
                         K=self.knn_k + 1, return_nn=True, return_sorted=True)
        self._knn_idx = res.idx[..., 1:]
        self._knn_dists = res.dists[..., 1:]
        self._knn_nn = res.knn[..., 1:, :]
        self.knn_gather = frnn.frnn_gather
        return self._knn_idx

    def denoise_normals(self, points, normals, num_points, **kwargs):
        """:500-526 -- weights exp(-((1 - <n, n_i>) / sigma)^2) exp(-|p - p_i|^2 P/2), cut at 32/P."""
        if points.shape[0] != 1:
            raise NotImplementedError("EdgeAwareProjection: one cloud per call (the reference's "
                                      "num_points / 2.0 broadcast is only valid for batch size 1, :515)")
        unit = F.normalize(normals, dim=-1)
        nbr_n = self.knn_gather(unit, self._knn_idx, num_points)                       # (1,P,K,3)
        sigma = self.sharpness_sigma = kwargs.get("sharpness_sigma", self.sharpness_sigma)
        bandwidth = num_points / 2.0                                                   # 1 / sigma_p^2 = P / 2 (:515)
        d2 = (self._knn_nn - points[:, :, None, :]).square().sum(dim=-1)
        w_space = torch.exp(-d2 * bandwidth).masked_fill(d2 > 16 / bandwidth, 0.0)
        w_angle = torch.exp(-(((1 - (nbr_n * unit[:, :, None, :]).sum(dim=-1)) / sigma) ** 2))
        w = w_space * w_angle
        mean = (nbr_n * w[..., None]).sum(dim=-2) / eps_denom(w.sum(dim=-1, keepdim=True))
        return F.normalize(mean, dim=-1), w_space, w_angle

    def upsample(self, points, n_points, model, num_points=None, **forward_kwargs):
        """:528-661.  points (1,P,3) on the level set; returns (points (1,P',3), num_points (1,))
        with P' = ceil(n_points * upsample_ratio); the new points are NOT yet projected (the
        driver does that, :429-434)."""
        upsample_ratio = forward_kwargs.pop("upsample_ratio", self.upsample_ratio)
        n_points = n_points * upsample_ratio
        n_points = n_points.ceil().long() if isinstance(n_points, torch.Tensor) else int(math.ceil(n_points))
        batch_size, P = points.shape[:2]
        if batch_size != 1:
            raise NotImplementedError("EdgeAwareProjection: one cloud per call (:515,:551)")
        if num_points is None:
            num_points = full_lengths(points)
        self._create_tree(points, refresh_tree=True, num_points_per_cloud=num_points)
        inv_sigma_spatial = num_points / 2.0
        spatial_dist = 16 / inv_sigma_spatial
        _, normals = self._compute_sdf_and_grad(points, model, **forward_kwargs)
        normals = F.normalize(normals, dim=-1, eps=1e-15)
        normals, _, _ = self.denoise_normals(points, normals, num_points)
        # LOP step (:559-596): data term along the filtered normal + density-weighted repulsion.
        dists, knn = self._knn_dists, self._knn_nn
        move_clip = dists[..., 0].mean().sqrt()
        diff = points[:, :, None, :] - knn
        zero = torch.zeros((), device=points.device)
        weight_lop = torch.exp(-torch.sum(normals[:, :, None, :] * diff, dim=-1) ** 2 * inv_sigma_spatial)
        weight_lop = torch.where(dists > spatial_dist, zero, weight_lop)
        spatial_w = torch.where(dists > spatial_dist, zero, torch.exp(-dists * inv_sigma_spatial))
        density_w = torch.sum(spatial_w, dim=-1) + 1.0
        move_data = torch.sum(weight_lop[..., None] * diff, dim=-2) / eps_denom(torch.sum(weight_lop, dim=-1, keepdim=True))
        move_repul = self.repulsion_mu * density_w[..., None] * torch.sum(spatial_w[..., None] * (knn - points[:, :, None, :]),
                                                                         dim=-2) / \
            eps_denom(torch.sum(spatial_w, dim=-1, keepdim=True))
        # (F.normalize without dim normalises over dim=1, the POINT axis: kept, :582-585)
        move_repul = F.normalize(move_repul) * move_repul.norm(dim=-1, keepdim=True).clamp_max(move_clip)
        move_data = F.normalize(move_data) * move_data.norm(dim=-1, keepdim=True).clamp_max(move_clip)
        points = points - (move_data + move_repul)
        n_remaining = n_points - num_points
        max_P = P // 10
        idx = self._knn_idx
        while True:
            if bool((n_remaining == 0).all()):
                break
            knn_pts = self.knn_gather(points, idx, num_points)
            knn_normals = self.knn_gather(normals, idx, num_points)
            # :609-633 -- sparsest candidate per point across normal discontinuities: one fused
            # O(K^2)-per-point kernel instead of the (1,P,K,K,3) tensor (11.5 kB/point at K = 31)
            father_sparsity = torch.empty(points.shape[:2], dtype=torch.float32, device=points.device)
            cand = torch.empty_like(points)
            args = [t.float().contiguous() for t in (points, normals, knn_pts, knn_normals)]
            _lib.call("iso_ear_candidates", *[_lib.ptr(t) for t in args], points.shape[1], idx.shape[-1],
                      float(self.edge_sensitivity), _lib.ptr(father_sparsity), _lib.ptr(cand), _lib.stream())
            sparsity_sorted = father_sparsity.sort(dim=1).indices[:, -max_P:]
            n_new_points = n_remaining.clamp(max=max_P)
            new_pts = torch.gather(cand, 1, sparsity_sorted.unsqueeze(-1).expand(-1, -1, 3))
            n_new, n_old = int(n_new_points[0]), int(num_points[0])
            points = torch.cat([new_pts[0][-n_new:], points[0, :n_old]], dim=0).unsqueeze(0)   # ([-0:] = all, as :642)
            n_remaining = n_remaining - n_new_points
            num_points = n_new_points + num_points
            self._create_tree(points, num_points_per_cloud=num_points, refresh_tree=True)
            idx = self._knn_idx
            _, normals = self._compute_sdf_and_grad(points, model, **forward_kwargs)
            normals = F.normalize(normals, dim=-1)
        return points, num_points


class SphereTracing(LevelSetProjection):
    """Sphere tracing along given rays (levelset_sampling.py:663-808): same constructor, same
    `project_points(ray0, ray_direction, model, latent=None)` and the same result dictionary.
    The whole loop -- value-only network evaluation, clamped advance, bounding-sphere test,
    compaction of the still-active rays -- runs in iso_trace_{sphere,siren,idr}; the reference's
    `max_points_per_pass` chunking is unnecessary (nothing is materialised per layer)."""

    def __init__(self, proj_max_iters=10, proj_tolerance=5e-5, max_points_per_pass=120000,
                 alpha=1.0, radius=1.0, padding=0.1, **kwargs):
        super().__init__(proj_max_iters=proj_max_iters, proj_tolerance=proj_tolerance,
                         max_points_per_pass=max_points_per_pass)
        self.alpha = alpha
        self.radius = radius
        self.padding = padding

    def _trace_packed(self, model, ray0, dirs):
        n, dev = ray0.shape[0], ray0.device
        out = torch.empty_like(ray0)
        sdf = torch.zeros((n,), dtype=torch.float32, device=dev)
        mask = torch.zeros((n,), dtype=torch.uint8, device=dev)
        p = _lib.ptr
        bound = float(self.padding + self.radius)
        T, tol, alpha = int(self.proj_max_iters), float(self.proj_tolerance), float(self.alpha)
        if getattr(model, "iso_analytic", None) == "sphere":
            c = [float(x) for x in model.center.tolist()]
            _lib.call("iso_trace_sphere", p(ray0), p(dirs), p(out), p(sdf), p(mask), n, c[0], c[1], c[2],
                      float(model.radius), alpha, bound, T, tol, _lib.stream())
        elif siren_spec(model) is not None:
            ps = PackedSiren(model, dev)
            ws = ps.workspace(n)
            _lib.call("iso_trace_siren", p(ray0), p(dirs), p(out), p(sdf), p(mask), n, p(ps.packed), ps.hidden,
                      ps.n_hidden, ps.omega_first, ps.omega_hidden, alpha, bound, T, tol, p(ws), ws.numel(),
                      _lib.stream())
            self._packed_cache = ps
        elif idr_spec(model) is not None:
            pk = PackedIdr(model, dev)
            ws = pk.workspace(n)
            _lib.call("iso_trace_idr", p(ray0), p(dirs), p(out), p(sdf), p(mask), n, p(pk.packed), pk.hidden,
                      pk.n_layers, pk.skip, pk.n_freq, 100.0, alpha, bound, T, tol, p(ws), ws.numel(),
                      _lib.stream())
            self._packed_cache = pk
        else:
            return self._trace_packed_generic(model, ray0, dirs)
        return out, sdf, mask.bool()

    def _trace_packed_generic(self, model, ray0, dirs):
        """The reference's loop (levelset_sampling.py:735-779) for models without a fused kernel."""
        cur = ray0.clone()
        n = cur.shape[0]
        active = torch.ones((n,), dtype=torch.bool, device=cur.device)
        inside = torch.ones((n,), dtype=torch.bool, device=cur.device)
        val = torch.zeros((n,), dtype=torch.float32, device=cur.device)
        trials = 0
        with torch.no_grad():
            model.eval()
            while True:
                val[active] = model.forward(cur[active]).sdf.reshape(-1)
                active = (val.abs() > 1e-1 * self.proj_tolerance) & inside
                if not (bool(active.any()) and trials < self.proj_max_iters):
                    break
                move = self.alpha * val[active].unsqueeze(-1) * dirs[active]
                move = F.normalize(move, dim=-1, eps=1e-15) * move.norm(dim=-1, keepdim=True).clamp_max(0.1)
                pts_active = cur[active] + move
                still = pts_active.norm(dim=-1) < (self.padding + self.radius)
                ins = inside.clone()
                ins[active] = still
                inside = ins
                cur[active & inside] = pts_active[still]
                trials += 1
        return cur, val, val.abs() <= self.proj_tolerance

    def project_points(self, ray0, ray_direction, model, latent=None, **forward_kwargs):
        """ray0, ray_direction (N,*,3) -> dict(levelset_points, network_eval_on_levelset_points,
        levelset_points_Dx, mask); `levelset_points_Dx` is the point tensor itself, as in the
        reference (:806)."""
        if latent is not None and latent.nelement() > 0:
            raise NotImplementedError("iso_points_amd: latent-conditioned networks are outside the hot path (c_dim=0)")
        if not ray0.is_cuda:
            raise RuntimeError("iso_points_amd: rays must be on the GPU; there is no CPU path")
        shp = ray0.shape
        r0 = ray0.detach().reshape(-1, 3).float().contiguous()
        rd = ray_direction.detach().reshape(-1, 3).float().contiguous()
        with torch.no_grad():
            if forward_kwargs:
                pts, val, mask = self._trace_packed_generic(
                    _KwModel(model, forward_kwargs), r0, rd)
            else:
                pts, val, mask = self._trace_packed(model, r0, rd)
        pts = pts.view(shp)
        return {"levelset_points": pts,
                "network_eval_on_levelset_points": val.view(shp[:-1]),
                "levelset_points_Dx": pts,
                "mask": mask.view(shp[:-1])}


def _value_gradient(network, points):
    """D_x F at `points` (..., 3), detached: the fused SDF+gradient kernels for SIREN / IDR-style
    networks, autograd for anything else.  (No model.eval() here: the sampling layers below run
    inside the training step.)"""
    flat = points.detach().reshape(-1, 3)
    if flat.is_cuda and siren_spec(network) is not None:
        from .sdf_models import siren_sdf_and_grad
        return siren_sdf_and_grad(network, flat.float().contiguous())[1].view(points.shape)
    if flat.is_cuda and idr_spec(network) is not None:
        from .sdf_models import idr_sdf_and_grad
        return idr_sdf_and_grad(network, flat.float().contiguous())[1].view(points.shape)
    with torch.enable_grad():
        x = points.detach().requires_grad_(True)
        f = network.forward(x).sdf
        (g,) = torch.autograd.grad([f], [x], torch.ones_like(f))
    return g.detach()


class SampleNetwork(nn.Module):
    """Eq. 13 (levelset_sampling.py:1170-1207): iso-points as a differentiable function of the
    network parameters, p - (F(p; theta) - F(p; theta_0)) D_xF^+ .  The value is p; what matters is
    d/d theta.  D_xF (theta-independent, detached in the reference too) comes from the fused
    value+gradient kernel instead of an autograd pass with retain_graph; F(p; theta) is the caller's
    module under autograd, since its graph IS the result."""

    def forward(self, network, levelset_points, return_eval=False):
        if not levelset_points.is_cuda:
            raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
        p = levelset_points.detach()
        grad = _value_gradient(network, p)                            # D_xF at theta_0, no graph
        value = network.forward(p).sdf                                # F(p; theta): the only term with a graph
        newton_dir = grad / eps_denom((grad * grad).sum(dim=-1, keepdim=True), 1e-17)
        moved = p - (value - value.detach()).view(p.shape[:-1] + (1,)) * newton_dir
        return (moved, value) if return_eval else moved


class DirectionalSamplingNetwork(SampleNetwork):
    """levelset_sampling.py:1370-1403: the same along a viewing ray -- depth
    t(theta) = t - (F - F_0) / (D_xF . v), point = cam_pos + t(theta) v."""

    def forward(self, network, iso_points, ray, cam_pos, c=None, return_eval=False):
        if not iso_points.is_cuda:
            raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
        p = iso_points.detach()
        grad = _value_gradient(network, p)
        value = network.forward(p).sdf
        view_dir = F.normalize(ray, dim=-1, p=2)
        depth = (p - cam_pos).norm(dim=-1, keepdim=True)
        slope = (grad * view_dir.detach()).sum(dim=-1, keepdim=True)  # D_xF . v
        moved = cam_pos + (depth - (value - value.detach()) / eps_denom(slope, 1e-10)) * view_dir
        return (moved, value) if return_eval else moved


class _KwModel(object):
    def __init__(self, model, kw):
        self.model, self.kw = model, kw

    def eval(self):
        self.model.eval()

    def forward(self, x):
        return self.model.forward(x, **self.kw)


def run_Secant_method(f_start, f_end, d_start, d_end, n_secant_steps, p0, ray_direction, decoder, c=None,
                      **forward_kwargs):
    """levelset_sampling.py:1331-1367.  `decoder` is an nn.Module (a FusedSdf is built for it) or
    a FusedSdf; every step is one value-only evaluation of the fused kernel."""
    sdf = decoder if isinstance(decoder, FusedSdf) else FusedSdf(decoder, p0.device)
    d_pred = -f_start * (d_end - d_start) / (f_end - f_start) + d_start
    for _ in range(n_secant_steps):
        p_mid = p0 + d_pred.unsqueeze(-1) * ray_direction
        f_mid = sdf(p_mid, **forward_kwargs)
        ind_start = torch.eq(torch.sign(f_mid), torch.sign(f_start))
        d_start = torch.where(ind_start, d_pred, d_start)
        f_start = torch.where(ind_start, f_mid, f_start)
        d_end = torch.where(ind_start, d_end, d_pred)
        f_end = torch.where(ind_start, f_end, f_mid)
        d_pred = -f_start * (d_end - d_start) / (f_end - f_start) + d_start
    return p0 + d_pred.unsqueeze(-1) * ray_direction


def find_zero_crossing_between_point_pairs(p0, p1, network, n_secant_steps=8, n_steps=100, is_occupancy=True,
                                           max_points=80000, c=None, allow_in_to_out=False, **forward_kwargs):
    """First sign change of the network value between point pairs + secant refinement
    (levelset_sampling.py:1210-1328).  p0, p1 (N,*,3) -> pt_pred (N,*,3), mask (N,*).
    The n_steps proposals per pair are evaluated by the value-only fused kernel in one call
    (`max_points` chunking is unnecessary); no host synchronisation before the secant loop:
    it runs over all pairs and the mask is applied at the end (the per-pair arithmetic is the same)."""
    if c is not None and c.nelement() > 0:
        raise NotImplementedError("iso_points_amd: latent-conditioned networks are outside the hot path (c_dim=0)")
    if not p0.is_cuda:
        raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
    sdf = network if isinstance(network, FusedSdf) else FusedSdf(network, p0.device)
    device, shp = p0.device, p0.shape
    p0 = p0.detach().reshape(-1, 3).float()
    p1 = p1.detach().reshape(-1, 3).float()
    n_pts = p0.shape[0]
    ray_direction = F.normalize(p1 - p0, p=2, dim=-1, eps=1e-10)
    d_proposal = torch.linspace(0, 1, steps=n_steps, device=device).view(1, n_steps) * \
        torch.norm(p1 - p0, p=2, dim=-1).unsqueeze(-1)
    p_proposal = p0.unsqueeze(-2) + ray_direction.unsqueeze(-2) * d_proposal.unsqueeze(-1)
    val = sdf(p_proposal.reshape(-1, 3), **forward_kwargs).view(n_pts, n_steps)
    compare = (lambda d: d < 0.0) if is_occupancy else (lambda d: d > 0.0)
    sign_matrix = torch.cat([torch.sign(val[..., :-1] * val[..., 1:]),
                             torch.ones(n_pts, 1, device=device)], dim=-1)
    cost_matrix = sign_matrix * torch.arange(n_steps, 0, -1, device=device).float()
    values, indices = torch.min(cost_matrix, -1)
    mask_sign_change = values < 0
    rows = torch.arange(n_pts, device=device)
    mask_out_to_in = compare(val[rows, indices])
    mask = mask_sign_change if allow_in_to_out else (mask_sign_change & mask_out_to_in)
    d_start, f_start = d_proposal[rows, indices], val[rows, indices]
    nxt = torch.clamp(indices + 1, max=n_steps - 1)
    d_end, f_end = d_proposal[rows, nxt], val[rows, nxt]
    p_pred = run_Secant_method(f_start, f_end, d_start, d_end, n_secant_steps, p0, ray_direction, sdf, None,
                               **forward_kwargs)
    pt_pred = torch.where(mask.unsqueeze(-1), p_pred, torch.ones_like(p_pred))
    return pt_pred.view(shp), mask.view(shp[:-1])


def mask_padded_to_list(values, mask):
    """DSS/utils/__init__.py:119-146 for padded inputs: per cloud, the rows where mask is True.  The number of True
    rows per cloud is read from the device once per mask tensor (true_counts); a cloud whose mask is all True is
    returned as it is (no boolean-index pass: nonzero + gather + a host read of its own, 0.1 ms at 1 M points) -- a VIEW
    of `values`, where the reference's boolean index returns a copy: clone it before mutating it in place."""
    n_true = true_counts(mask)
    P = int(mask.shape[1]) if mask.ndim > 1 else 0
    return [values[b] if n_true[b] == P else values[b][mask[b]] for b in range(values.shape[0])]


def sample_uniform_iso_points(model, n_points, init_points=None, bounding_sphere_radius=1.0, generator=None,
                              device=None):
    """levelset_sampling.py:1405-1445: project 4n random points -> drop |p| >= R -> wlop ->
    project + upsample(K=31) -> upsample(n) -> project.  Returns the iso-points (1, n', 3)
    (the reference wraps them in a pytorch3d Pointclouds)."""
    from .point_processing import upsample, wlop
    projector = UniformProjection(max_points_per_pass=16000, proj_max_iters=10, proj_tolerance=5e-5, knn_k=8)
    if init_points is None:
        if device is None:
            device = next(model.parameters()).device if any(True for _ in model.parameters()) else torch.device("cuda")
        init_points = ((torch.rand((1, n_points * 4, 3), generator=generator) - 0.5) * 2 * bounding_sphere_radius)
        init_points = init_points.to(device)
    res = projector.project_points(init_points, model, skip_resampling=True, skip_upsampling=True)
    boundary_mask = res["levelset_points"].norm(dim=-1) < bounding_sphere_radius
    pts = mask_padded_to_list(res["levelset_points"], res["mask"] & boundary_mask.view_as(res["mask"]))[0]
    pcl = pts.view(1, -1, 3).contiguous()
    X, num_X = wlop(pcl, None, min(0.5, n_points / max(pcl.shape[1], 1)), generator=generator)
    res = projector.project_points(_ClipLengths(X, num_X), model, skip_resampling=True, skip_upsampling=False)
    pts = mask_padded_to_list(res["levelset_points"], res["mask"])[0].view(1, -1, 3).contiguous()
    up, num_up = upsample(pts, n_points)
    res = projector.project_points(_ClipLengths(up, num_up), model, skip_resampling=True, skip_upsampling=False)
    return mask_padded_to_list(res["levelset_points"], res["mask"])[0].view(1, -1, 3)


class _ClipLengths(object):
    """Minimal Pointclouds stand-in: padded points + per-cloud lengths."""

    def __init__(self, points_padded, num_points):
        self._p, self._n = points_padded, num_points

    def points_padded(self):
        return self._p

    def num_points_per_cloud(self):
        return self._n


def _c_ptr(t):
    """Device pointer of a possibly strided view (caller passes the strides separately)."""
    import ctypes
    if not t.is_cuda:
        raise RuntimeError("iso_points_amd: tensor must live on the GPU")
    return ctypes.c_void_p(t.data_ptr())
