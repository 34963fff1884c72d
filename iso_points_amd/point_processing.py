"""Host-side mirror of the iso-point parts of DSS/utils/point_processing.py
(SURVEY 8(a) rows a8/a9 and 8(f)):

  upsample            point_processing.py:281-362   sparsest-edge midpoint insertion
  wlop                point_processing.py:35-122    FPS subsample + 3 LOP iterations
  farthest_sampling   point_processing.py:473-499   (torch_cluster.fps)
  knn_points          pytorch3d.ops.knn_points as upsample uses it (:315,:358)
  denoise_normals     point_processing.py:241-278   bilateral normal filter on the FRNN neighbourhood

Inputs are padded tensors (N,P,3) + lengths (pytorch3d's Pointclouds container is out of scope;
objects exposing points_padded()/num_points_per_cloud() are accepted).  Neighbour search, the
K^2 sparsity scan and FPS run in libisopoints_hip.so; the remaining glue is a handful of torch
ops on the GPU, exactly where the reference has them.
"""
import math
from collections import namedtuple

import torch

from . import _lib
from . import frnn
from .levelset_sampling import (cloud_diag, convert_pointclouds_to_tensor, eps_denom, full_lengths, host_lengths,
                                with_host_lengths)

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, return_sorted=True):
    """Exact K nearest neighbours (squared distances, ascending; ties -> lower index).
    Implemented as the FRNN ring search with an infinite radius.  Like pytorch3d, slots that
    cannot be filled (cloud 2 has fewer than K points) hold idx 0 / dist 0."""
    N = p1.shape[0]
    r = torch.full((N,), float("inf"), dtype=torch.float32, device=p2.device)
    # cells sized so that the K-th neighbour is (mostly) closer than one cell: the query then ends
    # in its 3x3x3 fast path instead of the ring walk (same result for any cell size)
    l2 = frnn._as_lengths(lengths2, N, p2.shape[1], p2.device)
    grid = frnn.build_grid(p2.detach().float().contiguous(), l2, r, points_per_cell=max(8.0, 0.75 * K))   # measured optimum (tools/ear_bench.py)
    dists, idxs, nn, _ = frnn.frnn_grid_points(p1, p2, lengths1, lengths2, K=K, r=r, grid=grid,
                                               return_nn=return_nn)
    pad = idxs < 0
    dists = torch.where(pad, torch.zeros_like(dists), dists)
    idxs = torch.where(pad, torch.zeros_like(idxs), idxs)
    if nn is not None:
        nn = nn  # frnn already zero-fills the padded rows
    return _KNN(dists=dists, idx=idxs, knn=nn)


def _upsample_candidates(points, knn):
    """(N,P,3), (N,P,K,3) -> sparsity (N,P), candidate (N,P,3)  [point_processing.py:326-339]."""
    N, P, K = knn.shape[0], knn.shape[1], knn.shape[2]
    pts = points.detach().float().contiguous()
    kn = knn.detach().float().contiguous()
    spars = torch.empty((N, P), dtype=torch.float32, device=pts.device)
    cand = torch.empty((N, P, 3), dtype=torch.float32, device=pts.device)
    _lib.call("iso_upsample_candidates", _lib.ptr(pts), _lib.ptr(kn), N * P, K, _lib.ptr(spars), _lib.ptr(cand),
              _lib.stream())
    return spars, cand


def upsample(pcl, n_points, num_points=None, neighborhood_size=16, knn_result=None):
    """Iteratively add midpoints in the sparsest regions until every cloud has n_points
    (point_processing.py:281-362).  Returns (points_padded, num_points)."""
    points, np_conv = convert_pointclouds_to_tensor(pcl)
    if num_points is None:
        num_points = np_conv
    knn_k = neighborhood_size
    dev = points.device
    if not torch.is_tensor(n_points):
        n_points = torch.full_like(num_points, int(n_points))
    n_points = n_points.to(dev)
    if int(num_points.sum()) == 0:
        return points, num_points
    n_remaining = (n_points - num_points).to(dtype=torch.long)
    if bool((n_remaining <= 0).all()):
        return points, num_points

    def _knn(pts, lens):
        r = knn_points(pts, pts, lens, lens, K=knn_k + 1, return_nn=True, return_sorted=True)
        return _KNN(dists=r.dists[..., 1:], idx=r.idx[..., 1:], knn=r.knn[..., 1:, :])

    if knn_result is None:
        knn_result = _knn(points, num_points)
    while True:
        if bool((n_remaining == 0).all()):
            break
        batch_size, P, _ = points.shape
        max_P = P // 8
        father_sparsity, cand = _upsample_candidates(points, knn_result.knn)            # :331-339
        sparsity_sorted = father_sparsity.sort(dim=1).indices
        n_new_points = n_remaining.clone()
        n_new_points[n_new_points > max_P] = max_P
        sparsity_sorted = sparsity_sorted[:, -max_P:] if max_P > 0 else sparsity_sorted[:, :0]
        new_pts = torch.gather(cand, 1, sparsity_sorted.unsqueeze(-1).expand(-1, -1, 3))
        lens = [int(x) for x in num_points.tolist()]
        nnew = [int(x) for x in n_new_points.tolist()]
        total = []
        for b in range(batch_size):
            # reference quirk kept on purpose: `new_pts[b][-n:]` with n == 0 is the WHOLE array
            # (point_processing.py:352), so a cloud that is already full in a heterogeneous batch
            # still gets max_P rows prepended (the reference warns about such batches, :305-307)
            nb = new_pts[b][new_pts.shape[1] - nnew[b]:] if nnew[b] > 0 else new_pts[b]
            total.append(torch.cat([nb, points[b, :lens[b]]], dim=0))                    # :350-353
        mx = max(t.shape[0] for t in total)
        points = points.new_zeros((batch_size, mx, 3))
        for b, t in enumerate(total):
            points[b, : t.shape[0]] = t
        n_remaining = n_remaining - n_new_points
        num_points = n_new_points + num_points
        if max_P == 0:
            break
        knn_result = _knn(points, num_points)
    return points, num_points


# ----------------------------------------------------------------------------- FPS / WLOP
def farthest_sampling(points, num_points, ratio, random_start=False, generator=None):
    """Farthest-point sampling of ceil(ratio * n) points per cloud (torch_cluster.fps semantics;
    point_processing.py:473-499).  points (N,P,3) padded.  Returns (sampled_padded, num_sampled,
    idx_padded).  random_start=False starts from point 0 of each cloud (deterministic); True draws
    the start like torch_cluster does."""
    N, P, _ = points.shape
    dev = points.device
    lens = host_lengths(num_points)
    ns = [int(math.ceil(ratio * l)) for l in lens]
    mx = max(ns) if ns else 0
    pts = points.detach().float().contiguous()
    start = torch.zeros((N,), dtype=torch.int64, device=dev)
    if random_start:
        u = torch.rand((N,), generator=generator, device="cpu")
        start = (u * torch.tensor(lens, dtype=torch.float32)).long().clamp(max=max(max(lens) - 1, 0)).to(dev)
    out_idx = torch.full((N, max(mx, 1)), -1, dtype=torch.int64, device=dev)
    nsamp = with_host_lengths(torch.tensor(ns, dtype=torch.int64, device=dev), ns)
    work = torch.empty((_lib.load().iso_farthest_point_sampling_work_floats(N, P),), dtype=torch.float32, device=dev)
    if mx > 0:
        _lib.call("iso_farthest_point_sampling", _lib.ptr(pts), _lib.ptr(num_points.to(torch.int64).contiguous()),
                  _lib.ptr(nsamp), _lib.ptr(start), N, P, out_idx.shape[1], _lib.ptr(work), _lib.ptr(out_idx),
                  _lib.stream())
    safe = out_idx.clamp(min=0)
    sampled = torch.gather(pts, 1, safe.unsqueeze(-1).expand(-1, -1, 3))
    sampled = sampled * (out_idx >= 0).unsqueeze(-1).float()
    return sampled[:, :mx], nsamp, out_idx[:, :mx]


def wlop(points, num_points=None, ratio=0.5, neighborhood_size=16, iters=3, repulsion_mu=0.5,
         generator=None, perturb=True, random_start=False):
    """Weighted locally optimal projection (point_processing.py:35-122) on padded tensors.
    Returns (X (N,I,3), num_points_X).  `perturb` adds the reference's randn*h*0.1 offset (:59)."""
    P, num_P = convert_pointclouds_to_tensor(points)
    if num_points is not None:
        num_P = num_points
    dev = P.device
    lensP = host_lengths(num_P)
    N = P.shape[0]
    # bbox diagonal per cloud over the valid points (Pointclouds.get_bounding_boxes, :43-45)
    diag = cloud_diag(P, num_P.to(torch.int64).contiguous())
    h = 4 * torch.sqrt(diag / num_P.float())
    search_radius = torch.clamp(h * neighborhood_size, max=0.2)          # min(h*ns, 0.2) per cloud (:47)
    theta_sigma_inv = 16 / h / h
    if ratio < 1.0:
        X, num_X, _ = farthest_sampling(P, num_P, ratio, random_start=random_start, generator=generator)
    elif ratio == 1.0:
        X, num_X = P.clone(), num_P
    else:
        raise ValueError("ratio must be less or equal to 1.0")
    if perturb:
        noise = torch.randn(X.shape, generator=generator, device="cpu").to(dev) if generator is not None \
            else torch.randn_like(X)
        X = X + noise * (h * 0.1).view(-1, 1, 1)
    falloff = theta_sigma_inv.view(-1, 1, 1)

    def weight(r2, ids):                       # theta(r) = exp(-16 r^2 / h^2), zero for a missing neighbour
        return torch.exp(-r2 * falloff).masked_fill(ids < 0, 0.0)

    K = neighborhood_size
    _, ids_pp, _, grid = frnn.frnn_grid_points(P, P, num_P, num_P, K=K + 1, r=search_radius)
    ids_pp = ids_pp[..., 1:].contiguous()
    gap_pp = torch.norm(P.unsqueeze(-2) - frnn.frnn_gather(P, ids_pp), dim=-1)
    dens_P = weight(gap_pp ** 2, ids_pp).sum(dim=-1) + 1                    # local density of the input cloud (:66-70)
    for _ in range(iters):
        _, ids_xp, _, grid = frnn.frnn_grid_points(X, P, num_X, num_P, K=K, r=search_radius, grid=grid)
        _, ids_xx, _, _ = frnn.frnn_grid_points(X, X, num_X, num_X, K=K + 1, r=search_radius)
        ids_xx = ids_xx[..., 1:].contiguous()
        anchor = frnn.frnn_gather(P, ids_xp)                                 # input points near each sample
        to_anchor = X.unsqueeze(-2) - anchor
        to_peer = X.unsqueeze(-2) - frnn.frnn_gather(X, ids_xx)
        r2_anchor, r2_peer = (to_anchor ** 2).sum(dim=-1), (to_peer ** 2).sum(dim=-1)
        dens_X = torch.exp(-r2_peer * falloff).sum(dim=-1) + 1
        # attraction to the input, each anchor discounted by its density; repulsion among the samples (:84-118)
        pull = weight(r2_anchor, ids_xp) / eps_denom(to_anchor.norm(dim=-1)) / \
            frnn.frnn_gather(dens_P.unsqueeze(-1), ids_xp).squeeze(-1)
        pull = pull.masked_fill(ids_xp < 0, 0.0)
        push = dens_X.unsqueeze(-1) * (torch.exp(-r2_peer * falloff) / eps_denom(to_peer.norm(dim=-1)))
        push = push.masked_fill(ids_xx < 0, 0.0)
        X = (pull[..., None] * anchor).sum(dim=-2) / eps_denom(pull.sum(dim=-1, keepdim=True)) + \
            repulsion_mu * (push[..., None] * to_peer).sum(dim=-2) / eps_denom(push.sum(dim=-1, keepdim=True))
    return X, num_X


def resample_uniformly(points, num_points=None, neighborhood_size=8, shrink_ratio=0.5, repulsion_mu=1.0,
                       generator=None):
    """wlop + upsample back to the original count (point_processing.py:126-166, tensor inputs)."""
    pts, num = convert_pointclouds_to_tensor(points)
    if num_points is not None:
        num = num_points
    X, num_X = wlop(pts, num, ratio=shrink_ratio, repulsion_mu=repulsion_mu, generator=generator)
    return upsample(X, num, num_points=num_X)


def denoise_normals(points, normals, sharpness_sigma=30, knn_result=None, neighborhood_size=16):
    """Bilateral normal filter (point_processing.py:241-278), one cloud: weights
    exp(-((1 - <n, n_i>) / sharpness_sigma)^2) exp(-|p - p_i|^2 P/2) cut at |p - p_i|^2 > 32/P, over
    the FRNN neighbourhood of radius min(4 sqrt(diag/P) K, 0.2).  `sharpness_sigma` divides as given
    (the reference does not convert the angle here, :263).  A `knn_result` must carry idx (and dists);
    its `knn` is gathered when missing (the reference's own branch for a given `knn` reads an
    unbound local, :269)."""
    points, num_points = convert_pointclouds_to_tensor(points)
    if not points.is_cuda:
        raise RuntimeError("iso_points_amd: points must be on the GPU; there is no CPU path")
    if points.shape[0] != 1:
        raise NotImplementedError("denoise_normals: one cloud per call (math.sqrt(diag / P) at :249 "
                                  "needs a single-element tensor)")
    unit = torch.nn.functional.normalize(normals, dim=-1)
    if knn_result is None:
        extent = (points.amax(dim=-2) - points.amin(dim=-2)).norm(dim=-1)
        radius = min(4 * neighborhood_size * math.sqrt(extent / points.shape[1]), 0.2)
        d2, ids, _, _ = frnn.frnn_grid_points(points, points, num_points, num_points, K=neighborhood_size + 1, r=radius,
                                              grid=None, return_nn=True)
        knn_result = _KNN(dists=d2[..., 1:], idx=ids[..., 1:], knn=None)
    nbr_p = knn_result.knn if knn_result.knn is not None else frnn.frnn_gather(points, knn_result.idx, num_points)
    nbr_n = frnn.frnn_gather(unit, knn_result.idx, num_points)
    bandwidth = num_points / 2.0                                      # 1 / sigma_p^2 = P / 2
    gap2 = (nbr_p - points[:, :, None, :]).square().sum(dim=-1)
    w = torch.exp(-gap2 * bandwidth).masked_fill(gap2 > 16 / bandwidth, 0.0) * \
        torch.exp(-(((1 - (nbr_n * unit[:, :, None, :]).sum(dim=-1)) / sharpness_sigma) ** 2))
    mean = (nbr_n * w[..., None]).sum(dim=-2) / eps_denom(w.sum(dim=-1, keepdim=True))
    return torch.nn.functional.normalize(mean, dim=-1).view_as(normals)
