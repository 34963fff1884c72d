"""CPU oracle for the iso-point hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (iso_points_amd) never does and fails loudly if
the HIP library is missing.

Each function restates one piece of the reference (yifita/iso-points) with the
same tensor operations in the same order, in float32 on the CPU, and cites the
file:line it follows.  Pinning status (see DESIGN.md "Oracle"):
  * projection / repulsion / resample / EWA per-point setup: pinned against the
    reference's own Python (imported in the build container with shims) through
    the fixtures in tests/golden/ made by tests/golden/make_golden.py.
  * splat forward / backward: the C restatement in oracle/oracle_splat.c is
    pinned against the reference's DSS/csrc/rasterize_points_cpu.cpp compiled
    as-is into oracle/_ref/ (oracle/Makefile).
  * FRNN neighbour search: PARITY UNPINNED -- lxxue/FRNN@eab337f is a
    third-party dependency whose source is not in the reference checkout; the
    contract below is restated from the call sites and checked against brute
    force only.
"""
from collections import namedtuple
import math

import numpy as np
import torch
import torch.nn.functional as F

ProjectionResult = namedtuple("ProjectionResult", ("points", "normals", "mask"))
SdfOut = namedtuple("SdfOut", ("sdf",))


# --------------------------------------------------------------------------- helpers
def eps_denom(denom, eps=1e-17):
    """DSS/utils/mathHelper.py:14-18 -- sign-preserving clamp of |x| to >= eps."""
    denom_sign = denom.sign() + (denom == 0.0).type_as(denom)
    return denom_sign * torch.clamp(denom.abs(), eps)


def eps_sqrt(squared, eps=1e-17):
    """DSS/utils/mathHelper.py:20-25."""
    return torch.clamp(squared.abs(), eps)


def first_idx_from_num(num_points):
    """DSS/utils/__init__.py:26-29."""
    return F.pad(num_points, (1, 0), "constant", 0).cumsum(0)[:-1]


def padded_to_packed_list(padded, num_points):
    return torch.cat([padded[b, : int(n)] for b, n in enumerate(num_points.tolist())], dim=0)


def packed_to_padded(packed, num_points, pad_value=0.0):
    B = len(num_points)
    mx = int(max(num_points.tolist())) if B else 0
    out = packed.new_full((B, mx) + tuple(packed.shape[1:]), pad_value)
    s = 0
    for b, n in enumerate(num_points.tolist()):
        out[b, :n] = packed[s : s + n]
        s += n
    return out


# --------------------------------------------------------------------------- SDF models
class SphereSDF(torch.nn.Module):
    """Analytic SDF |x - c| - R with the model contract of DSS/models/common.py:21-23
    (forward returns an object with `.sdf` of shape (M,1))."""

    def __init__(self, center=(0.0, 0.0, 0.0), radius=1.0):
        super().__init__()
        self.register_buffer("center", torch.tensor(center, dtype=torch.float32))
        self.radius = float(radius)

    def forward(self, x, **kwargs):
        return SdfOut(sdf=(x - self.center).norm(dim=-1, keepdim=True) - self.radius)


class SirenSDF(torch.nn.Module):
    """SIREN SDF, a restatement of Siren/SineLayer (DSS/models/common.py:56-165) with
    c_dim=0 and a linear head: h0=sin(w0(W0x+b0)), hi=sin(w(Wih+bi)), out=WLh+bL.
    Same initialisation (common.py:77-84, :128-131)."""

    def __init__(self, dim=3, hidden_size=256, n_layers=3, first_omega_0=30.0, hidden_omega_0=30.0):
        super().__init__()
        self.hidden_size, self.n_layers = hidden_size, n_layers
        self.first_omega_0, self.hidden_omega_0 = float(first_omega_0), float(hidden_omega_0)
        lins = [torch.nn.Linear(dim, hidden_size)]
        with torch.no_grad():
            lins[0].weight.uniform_(-1 / dim, 1 / dim)
        for _ in range(n_layers):
            lin = torch.nn.Linear(hidden_size, hidden_size)
            with torch.no_grad():
                b = np.sqrt(6 / hidden_size) / hidden_omega_0
                lin.weight.uniform_(-b, b)
            lins.append(lin)
        head = torch.nn.Linear(hidden_size, 1)
        with torch.no_grad():
            b = np.sqrt(6 / hidden_size) / hidden_omega_0
            head.weight.uniform_(-b, b)
        lins.append(head)
        self.lins = torch.nn.ModuleList(lins)

    def forward(self, x, **kwargs):
        h = torch.sin(self.first_omega_0 * self.lins[0](x))
        for lin in self.lins[1:-1]:
            h = torch.sin(self.hidden_omega_0 * lin(h))
        return SdfOut(sdf=self.lins[-1](h))

    def raw_weights(self):
        """Packed f32 buffer in the order include/isopoints.h documents."""
        parts = []
        for lin in self.lins:
            parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
        return torch.cat(parts).float().contiguous()


def fit_siren_to_sphere(model, radius=1.0, steps=300, seed=0, lr=1e-4, batch=4096):
    """Short Adam fit of a SirenSDF to |x|-R so Newton projection converges
    (SURVEY 8(d) cfg 2: 'weights fitted to the sphere for convergence realism')."""
    g = torch.Generator().manual_seed(seed)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    for _ in range(steps):
        x = (torch.rand(batch, 3, generator=g) - 0.5) * 3.0
        y = x.norm(dim=-1, keepdim=True) - radius
        loss = ((model(x).sdf - y) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    return model


def compute_sdf_and_grad(points, model, max_points_per_pass=120000, **kw):
    """UniformProjection._compute_sdf_and_grad, levelset_sampling.py:142-170
    (`.detach()` instead of `.detach_()` on the split view: torch>=2 rejects the
    in-place form)."""
    shp = points.shape
    packed = points.reshape(-1, 3)
    grads, evals = [], []
    with torch.no_grad():
        model.eval()
        for sub in torch.split(packed, max_points_per_pass, dim=0):
            with torch.enable_grad():
                x = sub.detach().requires_grad_(True)
                out = model.forward(x, **kw).sdf
                (g,) = torch.autograd.grad([out], [x], torch.ones_like(out), retain_graph=False)
            grads.append(g)
            evals.append(out.detach())
        if len(grads) == 0:
            return packed.new_zeros(shp[:-1]), packed.new_zeros(shp)
        return torch.cat(evals, 0).view(shp[:-1]), torch.cat(grads, 0).view(shp)


# --------------------------------------------------------------------------- A. projection
def project_points(model, points, num_points, proj_max_iters=10, proj_tolerance=5e-5,
                   max_points_per_pass=120000, **kw):
    """UniformProjection._project_points, levelset_sampling.py:290-351.
    points (B,P,3) padded, num_points (B,) long -> ProjectionResult (padded)."""
    points_packed = padded_to_packed_list(points, num_points).clone()
    not_converged = torch.ones(points_packed.shape[0], dtype=torch.bool)
    normals_packed = torch.zeros_like(points_packed)
    it = 0
    while True:
        curr_points = points_packed[not_converged]
        curr_sdf, curr_grad = compute_sdf_and_grad(curr_points, model, max_points_per_pass, **kw)
        normals_packed[not_converged] = curr_grad
        curr_not_converged = curr_sdf.reshape(-1).abs() > proj_tolerance
        nc = not_converged.clone()
        nc[not_converged] = curr_not_converged          # :328 (self-indexed write, torch>=2 safe)
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
    valid = ~not_converged
    return ProjectionResult(packed_to_padded(points_packed, num_points),
                            packed_to_padded(normals_packed, num_points),
                            packed_to_padded(valid.view(-1, 1).float(), num_points).squeeze(-1).bool())


# --------------------------------------------------------------------------- B. FRNN (contract)
def _d2(q, p):
    """(dx*dx + dy*dy) + dz*dz in float32, no fused multiply-add."""
    d = (q[:, None, :] - p[None, :, :]).astype(np.float32)
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=8, r=0.1,
                     return_nn=False, use_tree=None):
    """Contract of frnn.frnn_grid_points as the reference uses it
    (levelset_sampling.py:132-138): for each query the K nearest points of cloud 2
    with squared distance < r^2, ascending by (d2, index), -1 padded.  Exact
    brute force; for big clouds a cKDTree only pre-selects candidates (radius
    r*(1+1e-4)), distances and order are still computed as above."""
    p1 = points1.detach().cpu().numpy().astype(np.float32)
    p2 = points2.detach().cpu().numpy().astype(np.float32)
    N, P1, _ = p1.shape
    P2 = p2.shape[1]
    l1 = [P1] * N if lengths1 is None else [int(x) for x in lengths1.tolist()]
    l2 = [P2] * N if lengths2 is None else [int(x) for x in lengths2.tolist()]
    rr = np.broadcast_to(np.asarray(r.detach().cpu().numpy() if torch.is_tensor(r) else r,
                                    dtype=np.float32).reshape(-1), (N,)) if N else np.zeros(0, np.float32)
    dists = np.full((N, P1, K), -1.0, np.float32)
    idxs = np.full((N, P1, K), -1, np.int64)
    for n in range(N):
        a, b = p1[n, : l1[n]], p2[n, : l2[n]]
        r2 = np.float32(rr[n]) * np.float32(rr[n])
        if len(a) == 0 or len(b) == 0:
            continue
        tree = use_tree if use_tree is not None else (len(a) * len(b) > 4e7)
        if not tree:
            step = max(1, int(2e7 // max(len(b), 1)))
            for s in range(0, len(a), step):
                d2 = _d2(a[s : s + step], b)
                for i in range(d2.shape[0]):
                    cand = np.nonzero(d2[i] < r2)[0]
                    if len(cand) == 0:
                        continue
                    order = np.lexsort((cand, d2[i, cand]))[:K]
                    sel = cand[order]
                    dists[n, s + i, : len(sel)] = d2[i, sel]
                    idxs[n, s + i, : len(sel)] = sel
        else:
            from scipy.spatial import cKDTree
            kd = cKDTree(b.astype(np.float64))
            lists = kd.query_ball_point(a.astype(np.float64), float(rr[n]) * (1 + 1e-4) + 1e-7)
            for i, cand in enumerate(lists):
                if len(cand) == 0:
                    continue
                cand = np.asarray(cand, dtype=np.int64)
                d = (a[i][None, :] - b[cand]).astype(np.float32)
                sq = d * d
                d2 = (sq[:, 0] + sq[:, 1]) + sq[:, 2]
                keep = d2 < r2
                cand, d2 = cand[keep], d2[keep]
                order = np.lexsort((cand, d2))[:K]
                dists[n, i, : len(order)] = d2[order]
                idxs[n, i, : len(order)] = cand[order]
    dists_t, idxs_t = torch.from_numpy(dists), torch.from_numpy(idxs)
    nn = frnn_gather(points2.detach().cpu().float(), idxs_t) if return_nn else None
    return dists_t, idxs_t, nn, None


def frnn_gather(x, idxs, lengths=None):
    """frnn.frnn_gather: x (N,P2,U), idxs (N,P1,K) -> (N,P1,K,U); zeros where idx<0."""
    N, P1, K = idxs.shape
    U = x.shape[-1]
    safe = idxs.clamp(min=0)
    out = torch.gather(x, 1, safe.reshape(N, P1 * K, 1).expand(-1, -1, U)).view(N, P1, K, U)
    return out * (idxs >= 0)[..., None].to(x.dtype)


def search_radius(points_padded, num_points, knn_k):
    """levelset_sampling.py:129-131.  Like the reference, min/max run over the
    whole padded tensor."""
    diag = (points_padded.max(dim=1).values - points_padded.min(dim=1).values).norm(dim=-1)
    return torch.sqrt(diag / num_points.float()) * knn_k


# --------------------------------------------------------------------------- C. repulsion / resample
def repulsion_step(points, normals_unit, idx, inv_sigma_spatial):
    """Body of UniformProjection.resample, levelset_sampling.py:268-284.
    points (1,P,3), normals_unit = F.normalize(normals_init), idx (1,P,K) long."""
    nn_normals = frnn_gather(normals_unit, idx)
    knn_nn = frnn_gather(points, idx)
    knn_diff = points[:, :, None, :] - knn_nn
    knn_dists = torch.sum(knn_diff ** 2, dim=-1)
    spatial_w = torch.exp(-knn_dists * inv_sigma_spatial)
    spatial_w[idx < 0] = 0
    density_w = torch.sum(spatial_w, dim=-1, keepdim=True) + 1.0
    pts_diff_proj = knn_diff - (knn_diff * nn_normals).sum(dim=-1, keepdim=True) * nn_normals
    move = density_w * torch.sum(spatial_w[..., None] * pts_diff_proj, dim=-2) / \
        eps_denom(torch.sum(spatial_w, dim=-1, keepdim=True))
    return points + move


def resample(model, points_init, normals_init, num_points, sample_iters=1, knn_k=8,
             proj_tolerance=5e-5, max_points_per_pass=120000, frnn_fn=None, **kw):
    """UniformProjection.resample, levelset_sampling.py:239-288 (single cloud)."""
    frnn_fn = frnn_fn or frnn_grid_points
    B = points_init.shape[0]
    if num_points is None:
        num_points = torch.full((B,), points_init.shape[1], dtype=torch.long)
    full = points_init.new_full(points_init.shape[:-1], True, dtype=torch.bool)
    if sample_iters == 0 or points_init.nelement() < 2 * (knn_k + 1):
        return ProjectionResult(points_init, normals_init, full)
    flat = points_init.view(-1, 3)
    diag = (flat.max(dim=0).values - flat.min(0).values).norm().item()
    inv_sigma_spatial = num_points / diag
    points = points_init
    normals = F.normalize(normals_init, dim=-1)
    result, idx = None, None
    for it in range(sample_iters):
        if it % 2 == 0:
            r = search_radius(points, num_points, knn_k)
            _, idxs, _, _ = frnn_fn(points, points, num_points, num_points, K=knn_k + 1, r=r)
            idx = idxs[..., 1:]
        points = repulsion_step(points, normals, idx, inv_sigma_spatial)
        result = project_points(model, points, num_points, proj_max_iters=3,
                                proj_tolerance=proj_tolerance,
                                max_points_per_pass=max_points_per_pass, **kw)
    return result


def reduce_mask_padded(padded, mask):
    """DSS/utils/__init__.py:149-169: keep masked rows per cloud, re-pad with 0."""
    lst = [padded[b][mask[b]] for b in range(padded.shape[0])]
    mx = max([x.shape[0] for x in lst]) if lst else 0
    out = padded.new_zeros((len(lst), mx) + tuple(padded.shape[2:]))
    for b, x in enumerate(lst):
        out[b, : x.shape[0]] = x
    return out


# --------------------------------------------------------------------------- upsample / insert / wlop / FPS
KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=True):
    """pytorch3d.ops.knn_points as upsample uses it (point_processing.py:315,358): exact K nearest,
    squared distances ascending; unfilled slots idx 0 / dist 0.  PARITY UNPINNED (pytorch3d is
    absent); brute force with the same (d2, index) order as the FRNN contract."""
    d, i, nn, _ = frnn_grid_points(p1, p2, lengths1, lengths2, K=K, r=float("inf"), return_nn=return_nn,
                                   use_tree=False)
    pad = i < 0
    d = torch.where(pad, torch.zeros_like(d), d)
    i = torch.where(pad, torch.zeros_like(i), i)
    return KNN(dists=d, idx=i, knn=nn)


def upsample(points, n_points, num_points=None, neighborhood_size=16):
    """point_processing.upsample, point_processing.py:281-362 (tensor inputs)."""
    if num_points is None:
        num_points = torch.full((points.shape[0],), points.shape[1], dtype=torch.long)
    knn_k = neighborhood_size
    if not torch.is_tensor(n_points):
        n_points = torch.full_like(num_points, int(n_points))
    if num_points.sum() == 0:
        return points, num_points
    n_remaining = (n_points - num_points).to(dtype=torch.long)
    if (n_remaining <= 0).all():
        return points, num_points

    def _knn(pts, lens):
        r = knn_points(pts, pts, lens, lens, K=knn_k + 1, return_nn=True)
        return KNN(dists=r.dists[..., 1:], idx=r.idx[..., 1:], knn=r.knn[..., 1:, :])

    knn_result = _knn(points, num_points)
    while True:
        if (n_remaining == 0).all():
            break
        sparse_pts, sparse_knn = points, knn_result.knn
        batch_size, P, _ = sparse_pts.shape
        max_P = P // 8
        mid_points = (sparse_knn + 2 * sparse_pts[..., None, :]) / 3
        mid_nn_diff = mid_points.unsqueeze(-2) - sparse_knn.unsqueeze(-3)
        min_dist2 = torch.norm(mid_nn_diff, dim=-1).min(dim=-1)[0]
        father_sparsity, father_nb = min_dist2.max(dim=-1)
        sparsity_sorted = father_sparsity.sort(dim=1).indices
        n_new_points = n_remaining.clone()
        n_new_points[n_new_points > max_P] = max_P
        sparsity_sorted = sparsity_sorted[:, -max_P:]
        sel = mid_points[torch.arange(mid_points.shape[0]).view(-1, 1, 1),
                         torch.arange(mid_points.shape[1]).view(1, -1, 1), father_nb.unsqueeze(-1)].squeeze(-2)
        new_pts = torch.gather(sel, 1, sparsity_sorted.unsqueeze(-1).expand(-1, -1, 3))
        total = []
        for b in range(batch_size):
            pts_b = points[b, : int(num_points[b])]
            total.append(torch.cat([new_pts[b][-int(n_new_points[b]):], pts_b], dim=0))
        mx = max(t.shape[0] for t in total)
        points = points.new_zeros((batch_size, mx, 3))
        for b, t in enumerate(total):
            points[b, : t.shape[0]] = t
        n_remaining = n_remaining - n_new_points
        num_points = n_new_points + num_points
        knn_result = _knn(points, num_points)
    return points, num_points


def insert(ref_points, ref_metrics, points, num_points):
    """UniformProjection.insert, levelset_sampling.py:172-233 (one reference cloud).
    Returns (child_pts padded, child_per_batch)."""
    batch_size = points.shape[0]
    diag = (points.view(-1, 3).max(dim=0).values - points.view(-1, 3).min(0).values).norm().item()
    avg_spacing = math.sqrt(diag / ref_points.shape[0])
    patch_size = 8
    knn_k = patch_size
    search_radius = min(avg_spacing * knn_k, 0.2)
    _, idxs, _, _ = frnn_grid_points(points, points, num_points, num_points, K=knn_k + 1, r=search_radius)
    cur_idx = idxs[..., 1:]
    metrics = ref_metrics
    num_ref = metrics.shape[0]
    threshold = min(metrics.median() * 2, metrics.max() * 0.5)
    ref_pts = ref_points[(metrics > threshold).squeeze(-1)].view(1, -1, 3)
    if ref_pts.shape[1] == 0 or ref_pts.shape[1] > min(50, int(num_ref / 20)):
        ref_pts = ref_points[metrics.sort(dim=0).indices[-max(min(50, int(num_ref / 20)), 1):, 0]].view(1, -1, 3)
    ref_b = ref_pts.expand(batch_size, -1, -1)
    dists_to_ref, _, _, _ = frnn_grid_points(points, ref_b, num_points, None, K=1, r=search_radius * 4)
    dists_to_ref = dists_to_ref.view(batch_size, -1)
    father_mask = (dists_to_ref < 4 * avg_spacing ** 2) & (dists_to_ref > 0)
    father_pts = points[father_mask]
    mother_pts = frnn_gather(points, cur_idx[..., -patch_size:])[father_mask]
    child_pts = (2 * father_pts.unsqueeze(-2) / 3 + mother_pts / 3).view(-1, 3)
    child_per_batch = father_mask.sum(-1) * mother_pts.shape[-2]
    return packed_to_padded(child_pts, child_per_batch), child_per_batch


def farthest_point_sampling(points, n_samples, start=0):
    """Exact FPS (torch_cluster.fps contract): squared f32 distances, ties -> lowest index."""
    p = points.detach().cpu().numpy().astype(np.float32)
    n = p.shape[0]
    mind = np.full((n,), np.finfo(np.float32).max, np.float32)
    out = np.zeros((n_samples,), np.int64)
    cur = int(start)
    for s in range(n_samples):
        out[s] = cur
        d = (p - p[cur]).astype(np.float32)
        sq = d * d
        d2 = (sq[:, 0] + sq[:, 1]) + sq[:, 2]
        mind = np.minimum(mind, d2)
        cur = int(np.argmax(mind))
    return torch.from_numpy(out)


def wlop_iterations(P, num_P, X, num_X, neighborhood_size=16, iters=3, repulsion_mu=0.5):
    """The LOP iterations of point_processing.wlop (point_processing.py:43-48,68-118) for a given
    start set X (= FPS subsample + perturbation in the reference)."""
    lo = torch.stack([P[b, : int(num_P[b])].min(dim=0).values for b in range(P.shape[0])])
    hi = torch.stack([P[b, : int(num_P[b])].max(dim=0).values for b in range(P.shape[0])])
    diag = torch.norm(lo - hi, dim=-1)
    h = 4 * torch.sqrt(diag / num_P.float())
    search_radius = torch.clamp(h * neighborhood_size, max=0.2)
    tsi = (16 / h / h).view(-1, 1, 1)

    def theta(r2):
        return torch.exp(-r2 * tsi)

    K = neighborhood_size
    _, idxs, _, _ = frnn_grid_points(P, P, num_P, num_P, K=K + 1, r=search_radius)
    idx_pp = idxs[..., 1:]
    deltapp = torch.norm(P.unsqueeze(-2) - frnn_gather(P, idx_pp), dim=-1)
    theta_pp = theta(deltapp ** 2)
    theta_pp[idx_pp < 0] = 0
    density_P = torch.sum(theta_pp, dim=-1) + 1
    for _ in range(iters):
        _, idx_xp, _, _ = frnn_grid_points(X, P, num_X, num_P, K=K, r=search_radius)
        _, idx_xx, _, _ = frnn_grid_points(X, X, num_X, num_X, K=K + 1, r=search_radius)
        idx_xx = idx_xx[..., 1:]
        nn_XtoP = frnn_gather(P, idx_xp)
        epsilon = X.unsqueeze(-2) - nn_XtoP
        delta = X.unsqueeze(-2) - frnn_gather(X, idx_xx)
        deltaxx2 = (delta ** 2).sum(dim=-1)
        deltaxp2 = (epsilon ** 2).sum(dim=-1)
        alpha = theta(deltaxp2) / eps_denom(epsilon.norm(dim=-1))
        beta = theta(deltaxx2) * torch.ones_like(deltaxx2) / eps_denom(delta.norm(dim=-1))
        density_X = torch.sum(theta(deltaxx2), dim=-1) + 1
        new_alpha = alpha / frnn_gather(density_P.unsqueeze(-1), idx_xp).squeeze(-1)
        new_alpha[idx_xp < 0] = 0
        new_beta = density_X.unsqueeze(-1) * beta
        new_beta[idx_xx < 0] = 0
        term_data = torch.sum(new_alpha[..., None] * nn_XtoP, dim=-2) / \
            eps_denom(torch.sum(new_alpha, dim=-1, keepdim=True))
        term_repul = repulsion_mu * torch.sum(new_beta[..., None] * delta, dim=-2) / \
            eps_denom(torch.sum(new_beta, dim=-1, keepdim=True))
        X = term_data + term_repul
    return X


# --------------------------------------------------------------------------- IDR-style SDF
class IdrSDF(torch.nn.Module):
    """Restatement of SDF (DSS/models/common.py:220-310): positional encoding (get_embedder
    :205-217, include_input, log-sampled 2^k, [sin, cos] per frequency), n_layers softplus(beta=100)
    layers with a skip concatenation /sqrt(2) into layer `skip_in`, geometric initialisation
    (:258-275), weight-norm as explicit (g, v) parameters (:277-278), final tanh (:305)."""

    def __init__(self, hidden_size=512, n_layers=8, bias=0.6, skip_in=(4,), num_frequencies=6):
        super().__init__()
        self.F = num_frequencies
        d0 = 3 + 6 * num_frequencies
        dims = [d0] + [hidden_size] * n_layers + [1]
        self.dims, self.skip_in, self.num_layers = dims, tuple(skip_in), len(dims)
        self.hidden_size, self.n_layers = hidden_size, n_layers
        self.v, self.g, self.b = torch.nn.ParameterList(), torch.nn.ParameterList(), torch.nn.ParameterList()
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = torch.nn.Linear(dims[l], out_dim)
            if l == self.num_layers - 2:
                torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                torch.nn.init.constant_(lin.bias, -bias)
            elif num_frequencies > 0 and l == 0:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
            elif num_frequencies > 0 and l in self.skip_in:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
            else:
                torch.nn.init.constant_(lin.bias, 0.0)
                torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            w = lin.weight.detach()
            self.v.append(torch.nn.Parameter(w.clone()))
            self.g.append(torch.nn.Parameter(w.norm(dim=1, keepdim=True).clone()))   # weight_norm init: g = |v|
            self.b.append(torch.nn.Parameter(lin.bias.detach().clone()))

    def embed(self, x):
        outs = [x]
        for k in range(self.F):
            f = 2.0 ** k
            outs += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(outs, -1)

    def weight(self, l):
        v = self.v[l]
        return v * (self.g[l] / v.norm(dim=1, keepdim=True))

    def forward(self, inp, **kwargs):
        inp = self.embed(inp) if self.F > 0 else inp
        x = inp
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                x = torch.cat([x, inp], -1) / np.sqrt(2)
            x = F.linear(x, self.weight(l), self.b[l])
            if l < self.num_layers - 2:
                x = F.softplus(x, beta=100)
        return SdfOut(sdf=torch.tanh(x))

    def raw_weights(self):
        parts = []
        for l in range(self.num_layers - 1):
            parts += [self.weight(l).detach().reshape(-1), self.b[l].detach().reshape(-1)]
        return torch.cat(parts).float().contiguous()


# --------------------------------------------------------------------------- G. ray queries
def sphere_trace(model, ray0, ray_direction, proj_max_iters=10, proj_tolerance=5e-5, alpha=1.0,
                 radius=1.0, padding=0.1):
    """SphereTracing.project_points, levelset_sampling.py:679-808, for one sub-batch (the
    max_points_per_pass split does not change per-ray results).  ray0, ray_direction (...,3) ->
    dict with the reference's keys (`levelset_points_Dx` is the point tensor, :806)."""
    shp = ray0.shape
    cur = ray0.reshape(-1, 3).clone().float()
    dirs = ray_direction.reshape(-1, 3).float()
    n = cur.shape[0]
    active = torch.ones(n, dtype=torch.bool)
    inside = torch.ones(n, dtype=torch.bool)
    val = torch.zeros(n, 1)
    trials = 0
    with torch.no_grad():
        while True:
            val[active] = model.forward(cur[active]).sdf.reshape(-1, 1)            # :741-760 (value only)
            active = (val.abs() > 1e-1 * proj_tolerance).squeeze(1) & inside        # :764-765
            if not (bool(active.any()) and trials < proj_max_iters):                # :768
                break
            pts_active = cur[active]
            move = alpha * val[active] * dirs[active]                               # :773-774
            move = F.normalize(move, dim=-1, eps=1e-15) * move.norm(dim=-1, keepdim=True).clamp_max(0.1)
            pts_active = pts_active + move
            still = pts_active.norm(dim=-1) < (padding + radius)                    # :778-779
            ins = inside.clone()
            ins[active] = still
            inside = ins
            cur[active & inside] = pts_active[still]                                # :780-781
            trials += 1
    valid = val.abs() <= proj_tolerance                                             # :790
    pts = cur.view(shp)
    return {"levelset_points": pts, "network_eval_on_levelset_points": val.view(shp[:-1]),
            "levelset_points_Dx": pts, "mask": valid.view(shp[:-1])}


def run_secant(f_start, f_end, d_start, d_end, n_secant_steps, p0, ray_direction, model):
    """run_Secant_method, levelset_sampling.py:1331-1367 (inputs are cloned, the reference updates
    them in place)."""
    f_start, f_end, d_start, d_end = f_start.clone(), f_end.clone(), d_start.clone(), d_end.clone()
    d_pred = -f_start * (d_end - d_start) / (f_end - f_start) + d_start
    for _ in range(n_secant_steps):
        p_mid = p0 + d_pred.unsqueeze(-1) * ray_direction
        with torch.no_grad():
            f_mid = model.forward(p_mid).sdf.squeeze(-1)
        ind_start = torch.eq(torch.sign(f_mid), torch.sign(f_start))
        d_start[ind_start] = d_pred[ind_start]
        f_start[ind_start] = f_mid[ind_start]
        d_end[~ind_start] = d_pred[~ind_start]
        f_end[~ind_start] = f_mid[~ind_start]
        d_pred = -f_start * (d_end - d_start) / (f_end - f_start) + d_start
    return p0 + d_pred.unsqueeze(-1) * ray_direction


def find_zero_crossing(p0, p1, model, n_secant_steps=8, n_steps=100, is_occupancy=True, allow_in_to_out=False):
    """find_zero_crossing_between_point_pairs, levelset_sampling.py:1210-1328 (c = None)."""
    shp = p0.shape
    p0 = p0.reshape(-1, 3).float()
    p1 = p1.reshape(-1, 3).float()
    n_pts = p0.shape[0]
    compare = (lambda d: d < 0.0) if is_occupancy else (lambda d: d > 0.0)
    ray_direction = F.normalize(p1 - p0, p=2, dim=-1, eps=1e-10)
    d_proposal = torch.linspace(0, 1, steps=n_steps).view(1, n_steps) * torch.norm(p1 - p0, p=2, dim=-1).unsqueeze(-1)
    p_proposal = p0.unsqueeze(-2) + ray_direction.unsqueeze(-2) * d_proposal.unsqueeze(-1)
    with torch.no_grad():
        val = model.forward(p_proposal.view(-1, 3)).sdf.view(n_pts, n_steps)
    sign_matrix = torch.cat([torch.sign(val[..., :-1] * val[..., 1:]), torch.ones(n_pts, 1)], dim=-1)
    cost_matrix = sign_matrix * torch.arange(n_steps, 0, -1).float()
    values, indices = torch.min(cost_matrix, -1)
    mask_sign_change = values < 0
    rows = torch.arange(n_pts)
    mask_out_to_in = compare(val[rows, indices])
    mask = mask_sign_change if allow_in_to_out else (mask_sign_change & mask_out_to_in)
    d_start = d_proposal[rows, indices][mask]
    f_start = val[rows, indices][mask]
    nxt = torch.clamp(indices + 1, max=n_steps - 1)
    d_end = d_proposal[rows, nxt][mask]
    f_end = val[rows, nxt][mask]
    p_pred = run_secant(f_start, f_end, d_start, d_end, n_secant_steps, p0[mask], ray_direction[mask], model)
    pt_pred = torch.ones(mask.shape + (3,))
    pt_pred[mask] = p_pred
    return pt_pred.view(shp), mask.view(shp[:-1])


def ray_nearest_point(ray0, cam_pos, points):
    """combined_modeling.py:340-345: dense (R,M) point-to-ray distances + topk(k=1, smallest)."""
    pC = points - cam_pos.view(1, 3)
    ray_sq = (pC[None, :, :] * ray0[:, None, :]).sum(-1) ** 2
    dist_to_ray = (pC ** 2).sum(-1).unsqueeze(0) - ray_sq
    d, nn_idx = torch.topk(dist_to_ray, k=1, dim=1, largest=False)
    return torch.gather(ray_sq, 1, nn_idx).view(-1), nn_idx.view(-1), d.view(-1), dist_to_ray


def insurface_segments(cam_pos, ray0, frontal_points, occluded_points):
    """combined_modeling.py:336-357 for one batch element."""
    sq1 = ray_nearest_point(ray0, cam_pos, occluded_points)[0]
    sq0 = ray_nearest_point(ray0, cam_pos, frontal_points)[0]
    valid = sq0 < sq1
    return eps_sqrt(sq0).sqrt(), eps_sqrt(sq1).sqrt(), valid


def lowest_sdf_on_segments(model, cam_pos, cam_ray, ray_len0, ray_len1, n_points_per_ray=64):
    """combined_modeling.py:366-386."""
    lin = torch.linspace(0, 1.0, n_points_per_ray + 2)[1:-1]
    lengths = lin * (ray_len1 - ray_len0).view(-1, 1) + ray_len0.view(-1, 1)
    cand = lengths.unsqueeze(-1) * cam_ray.unsqueeze(-2) + cam_pos.reshape(-1, 1, 3)
    with torch.no_grad():
        val = model.forward(cand.view(-1, 3)).sdf.view(-1, n_points_per_ray)
    p_idx = torch.argmin(val, dim=-1, keepdim=True)
    return torch.gather(cand, -2, p_idx.unsqueeze(-1).expand(-1, -1, 3)).squeeze(-2), val


# --------------------------------------------------------------------------- H. IDR ray tracing
def sphere_entry_exit(cam_pos, cam_rays, radius=1.0):
    """intersection_with_unit_sphere, DSS/utils/__init__.py:484-545.  cam_pos (B,1,3) or (B,3),
    cam_rays (B,R,3) unit -> entry (B,R,3), exit (B,R,3), hit (B,R).  Rays that miss get the
    tangent-plane depths (cam_dist -+ radius) / (-p.q / cam_dist)  (:533-541)."""
    q = cam_rays.reshape(cam_rays.shape[0], -1, 3)
    p = cam_pos.reshape(cam_pos.shape[0], 1, 3)
    pq = (p * q).sum(-1)                                        # :506
    foot = p - pq[..., None] * q                                # :508  closest point of the line to 0
    dist = foot.norm(p=2, dim=-1)
    cam_dist = p.norm(dim=-1)                                   # (B,1)
    hit = dist <= radius                                        # :511
    chord = torch.full_like(dist, 10.0)                         # :517-518
    chord[hit] = 2 * torch.sqrt((radius ** 2 - dist ** 2)[hit])
    z0 = torch.zeros_like(dist)
    z0[hit] = torch.sqrt(cam_dist.expand_as(dist)[hit] ** 2 - dist[hit] ** 2) - chord[hit] / 2.0    # :528-530
    z0[~hit] = ((cam_dist - radius) / eps_denom(-pq / cam_dist))[~hit]                                # :533-534
    entry = z0.unsqueeze(-1) * q + p
    exit_ = chord[..., None] * q + entry                        # :537-538
    z_far = ((radius + cam_dist) / eps_denom(-pq / cam_dist))[~hit]
    exit_[~hit] = z_far.unsqueeze(-1) * q[~hit] + p.expand_as(q)[~hit]
    return entry, exit_, hit


def _rt_two_ended_trace(sdf, cam, dirs, hit, z_in, z_out, thr, max_iters, line_step, line_iters):
    """RayTracing.sphere_tracing, levelset_sampling.py:920-1032, on flattened rays: cam, dirs (R,3),
    hit (R,), z_in / z_out (R,) depths of the bounding-sphere entry / exit."""
    R = dirs.shape[0]
    live = [hit.clone(), hit.clone()]                           # 0: marching forward from the entry, 1: back from the exit
    z = [torch.where(hit, z_in, torch.zeros(R)), torch.where(hit, z_out, torch.zeros(R))]      # :933-944
    pts = [torch.zeros(R, 3), torch.zeros(R, 3)]
    for e in (0, 1):
        pts[e][hit] = (cam + z[e].unsqueeze(-1) * dirs)[hit]
    z_min, z_max = z[0].clone(), z[1].clone()                   # :947-948
    step_sign = (1.0, -1.0)

    def masked_eval(e, m, into=None):
        out = torch.zeros(R) if into is None else into
        if bool(m.any()):
            out[m] = sdf(pts[e][m])
        return out

    nxt = [masked_eval(e, live[e]) for e in (0, 1)]             # :953-959
    iters = 0
    while True:
        cur = []
        for e in (0, 1):                                        # :963-975
            c = torch.zeros(R)
            c[live[e]] = nxt[e][live[e]]
            c[c <= thr] = 0
            cur.append(c)
            live[e] = live[e] & (c > thr)
        if (not bool(live[0].any()) and not bool(live[1].any())) or iters == max_iters:   # :977
            break
        iters += 1
        for e in (0, 1):                                        # :983-990
            z[e] = z[e] + step_sign[e] * cur[e]
            pts[e] = cam + z[e].unsqueeze(-1) * dirs
            nxt[e] = masked_eval(e, live[e])                    # :993-999
        over = [nxt[0] < 0, nxt[1] < 0]                         # :1001-1002
        k = 0
        while (bool(over[0].any()) or bool(over[1].any())) and k < line_iters:            # :1004
            back = (1 - line_step) / (2 ** k)
            for e in (0, 1):                                    # :1006-1021
                m = over[e]
                z[e][m] = z[e][m] - step_sign[e] * (back * cur[e][m])
                pts[e][m] = (cam + z[e].unsqueeze(-1) * dirs)[m]
                nxt[e] = masked_eval(e, m, into=nxt[e])
            over = [nxt[0] < 0, nxt[1] < 0]
            k += 1
        ordered = z[0] < z[1]                                   # :1027-1030
        live = [live[0] & ordered, live[1] & ordered]
    return pts[0], live[0], z[0], z[1], z_min, z_max


def _rt_secant(sdf, f_lo, f_hi, z_lo, z_hi, cam, dirs, n_secant_steps):
    """RayTracing.secant, levelset_sampling.py:1114-1133 (f_lo > 0 outside, f_hi < 0 inside)."""
    f_lo, f_hi, z_lo, z_hi = f_lo.clone(), f_hi.clone(), z_lo.clone(), z_hi.clone()
    z = -f_lo * (z_hi - z_lo) / (f_hi - f_lo) + z_lo
    for _ in range(n_secant_steps):
        f_mid = sdf(cam + z.unsqueeze(-1) * dirs)
        pos, neg = f_mid > 0, f_mid < 0
        z_lo[pos], f_lo[pos] = z[pos], f_mid[pos]
        z_hi[neg], f_hi[neg] = z[neg], f_mid[neg]
        z = -f_lo * (z_hi - z_lo) / (f_hi - f_lo) + z_lo
    return z


def _rt_ray_sampler(sdf, cam, dirs, object_mask, z_lo, z_hi, todo, n_steps, n_secant_steps, training):
    """RayTracing.ray_sampler, levelset_sampling.py:1034-1112: n_steps uniform samples between
    z_lo and z_hi on the `todo` rays, first sign change -> secant; rays without one (or, when
    training, outside the ground-truth mask) take the sample of lowest value."""
    R = dirs.shape[0]
    out_pts, out_z = torch.zeros(R, 3), torch.zeros(R)
    rows = torch.nonzero(todo, as_tuple=False).flatten()
    n = rows.numel()
    lin = torch.linspace(0, 1, steps=n_steps).view(1, -1)
    zs = (z_lo.unsqueeze(-1) + lin * (z_hi - z_lo).unsqueeze(-1))[todo]                  # (n, n_steps)  :1045-1046
    P = cam[todo].unsqueeze(1) + zs.unsqueeze(-1) * dirs[todo].unsqueeze(1)
    val = torch.cat([sdf(c) for c in torch.split(P.reshape(-1, 3), 80000, dim=0)]).reshape(n, n_steps)
    first_neg = torch.argmin(torch.sign(val) * torch.arange(n_steps, 0, -1).float().view(1, -1), -1)     # :1061-1063
    ar = torch.arange(n)
    out_pts[rows] = P[ar, first_neg]
    out_z[rows] = zs[ar, first_neg]
    in_gt = object_mask[todo]
    in_net = val[ar, first_neg] < 0                                                        # :1070-1071
    lowest = ~(in_gt & in_net)                                                             # :1074
    if bool(lowest.any()):
        j = torch.argmin(val[lowest], -1)
        out_pts[rows[lowest]] = P[lowest][torch.arange(j.numel()), j]
        out_z[rows[lowest]] = zs[lowest][torch.arange(j.numel()), j]
    net_mask = todo.clone()                                                                # :1084-1085
    net_mask[rows[~in_net]] = False
    sec = (in_net & in_gt) if training else in_net                                         # :1088
    if bool(sec.any()):
        k = first_neg[sec]
        m = torch.arange(k.numel())
        z = _rt_secant(sdf, val[sec][m, k - 1], val[sec][m, k], zs[sec][m, k - 1], zs[sec][m, k],
                       cam[rows[sec]], dirs[rows[sec]], n_secant_steps)
        out_pts[rows[sec]] = cam[rows[sec]] + z.unsqueeze(-1) * dirs[rows[sec]]
        out_z[rows[sec]] = z
    return out_pts, net_mask, out_z


def _rt_minimal_sdf(sdf, cam, dirs, sel, z_min, z_max, n_steps, uniform_steps):
    """RayTracing.minimal_sdf_points, levelset_sampling.py:1135-1167: the lowest of n_steps
    samples at depths drawn once (shared by all rays) from U(0,1), scaled to [z_min, z_max]."""
    u = torch.empty(n_steps).uniform_(0.0, 1.0) if uniform_steps is None else uniform_steps
    lo, hi = z_min[sel].unsqueeze(-1), z_max[sel].unsqueeze(-1)
    zs = u.unsqueeze(0).repeat(lo.shape[0], 1) * (hi - lo) + lo
    P = cam[sel].unsqueeze(1).repeat(1, n_steps, 1) + zs.unsqueeze(-1) * dirs[sel].unsqueeze(1).repeat(1, n_steps, 1)
    val = torch.cat([sdf(c) for c in torch.split(P.reshape(-1, 3), 100000, dim=0)]).reshape(-1, n_steps)
    j = val.min(-1)[1]
    ar = torch.arange(j.numel())
    return P[ar, j], zs[ar, j]


def ray_tracing(sdf, cam_loc, object_mask, ray_directions, training=False, object_bounding_sphere=1.0,
                sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1, sphere_tracing_iters=10,
                n_steps=100, n_secant_steps=8, uniform_steps=None):
    """RayTracing.forward, levelset_sampling.py:831-918.  sdf: (M,3) -> (M,); cam_loc (B,3);
    object_mask (B*R,) bool; ray_directions (B,R,3) unit.  Returns points (B*R,3), network mask
    (B*R,), depth (B*R,).  `uniform_steps` replaces the U(0,1) draw of :1142 (training only)."""
    B, R, _ = ray_directions.shape
    with torch.no_grad():
        entry, exit_, hit = sphere_entry_exit(cam_loc, ray_directions, radius=object_bounding_sphere)
        span = (torch.stack([entry, exit_], dim=-2) - cam_loc.view(B, 1, 3).unsqueeze(-2)).norm(dim=-1) \
            / ray_directions.unsqueeze(-2).norm(dim=-1)                                    # :846-847
        cam = cam_loc.unsqueeze(1).repeat(1, R, 1).reshape(-1, 3)
        dirs = ray_directions.reshape(-1, 3)
        hit = hit.reshape(-1)
        span = span.reshape(-1, 2)
        pts, todo, z0, z1, z_min, z_max = _rt_two_ended_trace(
            sdf, cam, dirs, hit, span[:, 0], span[:, 1], sdf_threshold, sphere_tracing_iters,
            line_search_step, line_step_iters)
        net_mask = z0 < z1                                                                 # :853
        if bool(todo.any()):                                                               # :856-875
            lo = torch.where(todo, z0, torch.zeros_like(z0))
            hi = torch.where(todo, z1, torch.zeros_like(z1))
            s_pts, s_mask, s_z = _rt_ray_sampler(sdf, cam, dirs, object_mask, lo, hi, todo, n_steps,
                                                 n_secant_steps, training)
            pts[todo] = s_pts[todo]
            z0[todo] = s_z[todo]
            net_mask[todo] = s_mask[todo]
        if not training:                                                                   # :883-886
            return pts, net_mask, z0
        in_mask = ~net_mask & object_mask & ~todo                                          # :892-894
        out_mask = ~object_mask & ~todo
        left_out = (in_mask | out_mask) & ~hit                                             # :896-904
        if bool(left_out.any()):
            z0[left_out] = -(dirs[left_out] * cam[left_out]).sum(-1)
            pts[left_out] = cam[left_out] + z0[left_out].unsqueeze(1) * dirs[left_out]
        sel = (in_mask | out_mask) & hit                                                   # :906-916
        if bool(sel.any()):
            z_min[net_mask & out_mask] = z0[net_mask & out_mask]
            m_pts, m_z = _rt_minimal_sdf(sdf, cam, dirs, sel, z_min, z_max, n_steps, uniform_steps)
            pts[sel] = m_pts
            z0[sel] = m_z
        return pts, net_mask, z0


# --------------------------------------------------------------------------- I. image look-up
def get_tensor_values(tensor, p, grid_sample=True, mode="bilinear", with_mask=False, squeeze_channel_dim=False):
    """get_tensor_values, DSS/utils/__init__.py:325-375, with grid_sample(padding_mode='reflection',
    align_corners=False) written out (ATen GridSampler: unnormalise, reflect about the pixel edges,
    clip, 4 corners nw/ne/sw/se).  tensor (B,C,H,W), p (B,N,2) -> (B,N,C)."""
    B, C, H, W = tensor.shape
    bi = torch.arange(B).view(B, 1)
    if not grid_sample:                                           # :357-363 (p is not modified here)
        x = ((p[..., 0] + 1) * (W - 1) / 2).long()
        y = ((p[..., 1] + 1) * (H - 1) / 2).long()
        values = tensor[bi, :, y, x]
    else:
        def src(c, size):
            x = ((c + 1) * size - 1) / 2
            x = (x + 0.5).abs()
            extra, flips = torch.fmod(x, float(size)), torch.floor(x / size)
            x = torch.where(flips % 2 == 0, extra - 0.5, (size - extra) - 0.5)
            return x.clamp(0, size - 1)
        x, y = src(p[..., 0].float(), W), src(p[..., 1].float(), H)
        if mode == "nearest":
            values = tensor[bi, :, torch.round(y).long(), torch.round(x).long()]
        else:
            x0, y0 = torch.floor(x), torch.floor(y)
            values = torch.zeros(B, p.shape[1], C)
            for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):      # nw, ne, sw, se
                wx = (x0 + 1 - x) if dx == 0 else (x - x0)
                wy = (y0 + 1 - y) if dy == 0 else (y - y0)
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                v = tensor[bi, :, yi.clamp(0, H - 1), xi.clamp(0, W - 1)]
                values = values + torch.where(ok.unsqueeze(-1), v * (wx * wy).unsqueeze(-1), torch.zeros(()))
    mask = torch.isfinite(values) & ~torch.isnan(values)          # valid_value_mask, :15-16
    if squeeze_channel_dim:
        values, mask = values.squeeze(-1), mask.squeeze(-1)
    return (values, mask) if with_mask else values


# --------------------------------------------------------------------------- J. differentiable samples
def _value_gradient(network, pts):
    with torch.enable_grad():
        x = pts.detach().requires_grad_(True)
        f = network.forward(x).sdf
        (g,) = torch.autograd.grad([f], [x], torch.ones_like(f))
    return g.detach()


def sample_network(network, levelset_points, return_eval=False):
    """SampleNetwork.forward, levelset_sampling.py:1175-1207 (Eq. 13): p - (F(p;theta) - F(p;theta0))
    D_xF / |D_xF|^2 -- the value is p, the derivative w.r.t. theta is that of the level set."""
    p = levelset_points.detach()
    Dx = _value_gradient(network, p)
    f = network.forward(p).sdf
    ssg = torch.sum(Dx ** 2, dim=-1, keepdim=True)
    out = p - (f - f.detach()).view(p.shape[:-1] + (1,)) * (Dx / eps_denom(ssg, 1e-17))
    return (out, f) if return_eval else out


def directional_sample(network, iso_points, ray, cam_pos, return_eval=False):
    """DirectionalSamplingNetwork.forward, levelset_sampling.py:1371-1403: the depth along the
    viewing ray as a function of theta, t - (F - F0) / (D_xF . v)."""
    p = iso_points.detach()
    Dx = _value_gradient(network, p)
    t = (p - cam_pos).norm(dim=-1, keepdim=True)
    f = network.forward(p).sdf
    ray = F.normalize(ray, dim=-1, p=2)
    along = torch.sum(Dx * ray.detach(), dim=-1, keepdim=True)
    t_theta = t - (f - f.detach()) / eps_denom(along, 1e-10)
    out = cam_pos + t_theta * ray
    return (out, f) if return_eval else out


# --------------------------------------------------------------------------- K. edge-aware resampling
class BoxSDF(torch.nn.Module):
    """Exact SDF of an axis-aligned box (half extents b): a shape WITH edges for the edge-aware
    tests.  Not from the reference -- test input only."""

    def __init__(self, half=(0.5, 0.4, 0.3)):
        super().__init__()
        self.register_buffer("half_extent", torch.tensor(half, dtype=torch.float32))

    def forward(self, x, **kwargs):
        q = x.abs() - self.half_extent
        out = q.clamp_min(0).norm(dim=-1) + q.max(dim=-1)[0].clamp_max(0)
        return SdfOut(out.unsqueeze(-1))


def knn_gather(x, idx, lengths=None):
    """pytorch3d.ops.knn_gather: x (N,M,U), idx (N,P,K) -> (N,P,K,U); slots beyond a cloud's
    length hold index 0 there."""
    N, P, K = idx.shape
    return torch.gather(x.unsqueeze(1).expand(N, P, x.shape[1], x.shape[2]), 2,
                        idx.clamp_min(0).unsqueeze(-1).expand(N, P, K, x.shape[2]))


def ear_tree(points, num_points, knn_k):
    """EdgeAwareProjection._create_tree, levelset_sampling.py:471-498: K+1 nearest, self dropped."""
    r = knn_points(points, points, num_points, num_points, K=knn_k + 1, return_nn=True)
    return KNN(dists=r.dists[..., 1:], idx=r.idx[..., 1:], knn=r.knn[..., 1:, :])


def ear_denoise_normals(points, normals, num_points, tree, sharpness_sigma):
    """EdgeAwareProjection.denoise_normals, levelset_sampling.py:500-526 (bilateral normal filter)."""
    normals = F.normalize(normals, dim=-1)
    knn_normals = knn_gather(normals, tree.idx, num_points)
    weights_n = torch.exp(-((1 - torch.sum(knn_normals * normals[:, :, None, :], dim=-1)) / sharpness_sigma) ** 2)
    inv_sigma = num_points / 2.0
    spatial_dist = 16 / inv_sigma
    deltap = tree.knn - points[:, :, None, :]
    deltap = torch.sum(deltap * deltap, dim=-1)
    weights_p = torch.exp(-deltap * inv_sigma)
    weights_p[deltap > spatial_dist] = 0
    weights = weights_p * weights_n
    out = torch.sum(knn_normals * weights[:, :, :, None], dim=-2) / eps_denom(torch.sum(weights, dim=-1, keepdim=True))
    return F.normalize(out, dim=-1), weights_p, weights_n


def ear_upsample(points, n_points, model, num_points=None, knn_k=31, repulsion_mu=0.5, sharpness_angle=15,
                 edge_sensitivity=1, upsample_ratio=1.5):
    """EdgeAwareProjection.upsample, levelset_sampling.py:528-661, one cloud (the reference's
    `num_points / 2.0` broadcasts are only valid for batch size 1).  Kept as written there,
    including F.normalize(move) over dim=1 -- the POINT axis -- at :582-585."""
    n_points = n_points * upsample_ratio                                                    # :535-540
    n_points = n_points.ceil().long() if torch.is_tensor(n_points) else int(math.ceil(n_points))
    B, P = points.shape[:2]
    assert B == 1
    if num_points is None:
        num_points = torch.full((B,), P, dtype=torch.long)
    sharp = 1 - math.cos(sharpness_angle / 180 * math.pi)
    tree = ear_tree(points, num_points, knn_k)                                            # :548
    inv_sigma = num_points / 2.0
    spatial_dist = 16 / inv_sigma
    _, normals = compute_sdf_and_grad(points, model)                                       # :555-557
    normals = F.normalize(normals, dim=-1, eps=1e-15)
    normals, _, _ = ear_denoise_normals(points, normals, num_points, tree, sharp)
    move_clip = tree.dists[..., 0].mean().sqrt()                                           # :568
    diff = points[:, :, None, :] - tree.knn
    weight_lop = torch.exp(-torch.sum(normals[:, :, None, :] * diff, dim=-1) ** 2 * inv_sigma)
    weight_lop[tree.dists > spatial_dist] = 0
    spatial_w = torch.exp(-tree.dists * inv_sigma)
    spatial_w[tree.dists > spatial_dist] = 0
    density_w = torch.sum(spatial_w, dim=-1) + 1.0
    move_data = torch.sum(weight_lop[..., None] * diff, dim=-2) / eps_denom(torch.sum(weight_lop, dim=-1, keepdim=True))
    move_repul = repulsion_mu * density_w[..., None] * torch.sum(spatial_w[..., None] * (-diff), dim=-2) / \
        eps_denom(torch.sum(spatial_w, dim=-1, keepdim=True))
    move_repul = F.normalize(move_repul) * move_repul.norm(dim=-1, keepdim=True).clamp_max(move_clip)
    move_data = F.normalize(move_data) * move_data.norm(dim=-1, keepdim=True).clamp_max(move_clip)
    points = points - (move_data + move_repul)                                             # :596
    n_remaining = n_points - num_points
    max_P = P // 10
    while not bool((n_remaining == 0).all()):                                              # :601-659
        knn_pts = knn_gather(points, tree.idx, num_points)
        knn_normals = knn_gather(normals, tree.idx, num_points)
        mid = (knn_pts + 2 * points[..., None, :]) / 3
        d = mid.unsqueeze(-2) - knn_pts.unsqueeze(-3)                                      # (1,P,K,K,3)
        edge = (2 - torch.sum(normals.unsqueeze(-2) * knn_normals, dim=-1)) ** edge_sensitivity
        m = torch.norm(d, dim=-1) - torch.sum((d * knn_normals.unsqueeze(-2)) ** 2, dim=-1)
        m = eps_sqrt(m.min(dim=-1)[0]).sqrt()
        sparsity, father_nb = (edge * m).max(dim=-1)
        order = sparsity.sort(dim=1).indices[:, -max_P:]
        n_new = n_remaining.clone()
        n_new[n_new > max_P] = max_P
        cand = mid[torch.arange(B), torch.arange(mid.shape[1]), father_nb]                 # :632 (B = 1)
        new_pts = torch.gather(cand, 1, order.unsqueeze(-1).expand(-1, -1, 3))
        points = torch.cat([new_pts[0][-int(n_new[0]):], points[0, :int(num_points[0])]], dim=0).unsqueeze(0)
        n_remaining = n_remaining - n_new
        num_points = n_new + num_points
        tree = ear_tree(points, num_points, knn_k)
        _, normals = compute_sdf_and_grad(points, model)
        normals = F.normalize(normals, dim=-1)
    return points, num_points


def ear_project_points(points, model, knn_k=31, sample_iters=5, proj_max_iters=10, **ear_kw):
    """LevelSetProjection driver (levelset_sampling.py:353-440) as EdgeAwareProjection runs it, one
    cloud, no ref_pcl: project, drop the unconverged, resample on the K-nearest tree, edge-aware
    upsample to ceil(P * ratio), project."""
    num_init = torch.tensor([points.shape[1]])
    r = project_points(model, points, num_init, proj_max_iters=proj_max_iters)
    keep = r.mask[0]
    pts, nrm = r.points[:, keep], r.normals[:, keep]
    num = keep.sum().view(1)

    def tree(p1, p2, l1, l2, K, r):
        t = knn_points(p1, p2, l1, l2, K=K, return_nn=True)
        return t.dists, t.idx, t.knn, None
    r = resample(model, pts, nrm, num, sample_iters=sample_iters, knn_k=knn_k, frnn_fn=tree)
    keep = r.mask[0]
    pts = r.points[:, keep]
    num = keep.sum().view(1)
    up, num = ear_upsample(pts, num_init, model, num, knn_k=knn_k, **ear_kw)
    return project_points(model, up, num, proj_max_iters=10)


def denoise_normals(points, normals, sharpness_sigma=30, neighborhood_size=16):
    """point_processing.denoise_normals, DSS/utils/point_processing.py:241-278 (knn_result=None, one
    cloud): FRNN neighbourhood of radius min(4 sqrt(diag/P) K, 0.2), bilateral weights."""
    num_points = torch.tensor([points.shape[1]])
    normals = F.normalize(normals, dim=-1)
    diag = (points.max(dim=-2)[0] - points.min(dim=-2)[0]).norm(dim=-1)
    r = min(4 * math.sqrt(diag / points.shape[1]) * neighborhood_size, 0.2)
    _, idxs, _, _ = frnn_grid_points(points, points, num_points, num_points, K=neighborhood_size + 1, r=r)
    idx = idxs[..., 1:]
    knn = frnn_gather(points, idx, num_points)
    knn_normals = frnn_gather(normals, idx, num_points)
    weights_n = torch.exp(-((1 - torch.sum(knn_normals * normals[:, :, None, :], dim=-1)) / sharpness_sigma) ** 2)
    inv_sigma = num_points / 2.0
    deltap = knn - points[:, :, None, :]
    deltap = torch.sum(deltap * deltap, dim=-1)
    weights_p = torch.exp(-deltap * inv_sigma)
    weights_p[deltap > 16 / inv_sigma] = 0
    weights = weights_p * weights_n
    out = torch.sum(knn_normals * weights[:, :, :, None], dim=-2) / eps_denom(torch.sum(weights, dim=-1, keepdim=True))
    return F.normalize(out, dim=-1).view_as(normals)
