"""IDR's two-ended ray tracer: `RayTracing` with the reference's constructor,
call signature and three results (DSS/models/levelset_sampling.py:810-1167).

Every network evaluation goes through the `sdf` callable the caller passes -- build it with
`sdf_models.FusedSdf(model, device)` to run the forward-only fused HIP kernels (SIREN split-fp16
MFMA, IDR x16).  What differs from the reference is the shape of the work handed to that callable:
the reference treats the two marching ends separately and compacts each with boolean masks before
every call (2 network calls, 2 masked scatters and several host syncs per iteration); here both
ends are one (2,R) state and each iteration makes ONE network call on the still-unfinished ends,
compacted into a list by the kernel that updates the state (its length is the loop condition the
reference reads with `.sum() == 0`, and the only host read).  The statements between two
network calls -- threshold, masks, advance, overshoot back-step, the next call's point list and
its length -- are one HIP kernel each (csrc/raymarch.hip: iso_raymarch_settle / _overshoot, and
iso_raymarch_secant for the false-position update).  Per ray the arithmetic is the
same statement for statement and the fused kernels' per-point results do not depend on which
other points share the launch, so the results do not depend on this regrouping.
The interval sampler and the minimal-value search only touch the (few) unfinished / mismatched
rays and keep the reference's compaction.
"""
import torch
import torch.nn as nn

from . import _lib
from .levelset_sampling import eps_denom


def sphere_entry_exit(cam_pos, cam_rays, radius=1.0):
    """intersection_with_unit_sphere (DSS/utils/__init__.py:484-545) without boolean indexing.
    cam_pos (B,3) / (B,1,3), cam_rays (B,R,3) unit -> entry, exit (B,R,3), hit (B,R).  Rays that
    miss the sphere get the depths of the two tangent planes orthogonal to the viewing axis."""
    q = cam_rays.reshape(cam_rays.shape[0], -1, 3)
    p = cam_pos.reshape(cam_pos.shape[0], 1, 3)
    pq = (p * q).sum(dim=-1)
    dist = torch.norm(p - pq[..., None] * q, p=2, dim=-1)          # line-to-centre distance (:508-509)
    cam_dist = torch.norm(p, dim=-1)
    hit = dist <= radius
    chord = torch.where(hit, 2 * torch.sqrt(radius ** 2 - dist ** 2), torch.full_like(dist, 10.0))
    axis = eps_denom(-pq / cam_dist)
    z_near = torch.where(hit, torch.sqrt(cam_dist ** 2 - dist ** 2) - chord / 2.0, (cam_dist - radius) / axis)
    entry = z_near.unsqueeze(-1) * q + p
    far_miss = ((radius + cam_dist) / axis).unsqueeze(-1) * q + p
    exit_ = torch.where(hit.unsqueeze(-1), chord[..., None] * q + entry, far_miss)
    return entry, exit_, hit


class RayTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5,
                 line_step_iters=1, sphere_tracing_iters=10, n_steps=100, n_secant_steps=8):
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_secant_steps = n_secant_steps

    # -- :831-918 --------------------------------------------------------------------------
    def forward(self, sdf, cam_loc, object_mask, ray_directions, uniform_steps=None):
        """sdf: (M,3) -> (M,) on the GPU; cam_loc (B,3); object_mask (B*R,) bool; ray_directions
        (B,R,3) unit.  Returns points (B*R,3), network_object_mask (B*R,), distances (B*R,).
        `uniform_steps` (n_steps,) overrides the U(0,1) depths of minimal_sdf_points (:1142,
        training only; by default drawn from the CPU generator as the reference does)."""
        if not ray_directions.is_cuda:
            raise RuntimeError("iso_points_amd: rays must be on the GPU; there is no CPU path")
        B, R, _ = ray_directions.shape
        with torch.no_grad():
            entry, exit_, hit = sphere_entry_exit(cam_loc, ray_directions, radius=self.object_bounding_sphere)
            span = (torch.stack([entry, exit_], dim=-2) - cam_loc.view(B, 1, 3).unsqueeze(-2)).norm(dim=-1) \
                / ray_directions.unsqueeze(-2).norm(dim=-1)
            cam = cam_loc.unsqueeze(1).expand(B, R, 3).reshape(-1, 3)
            dirs = ray_directions.reshape(-1, 3)
            hit = hit.reshape(-1)
            object_mask = object_mask.reshape(-1)
            pts, todo, z, z_min, z_max = self.sphere_tracing(sdf, cam, dirs, hit, span.reshape(-1, 2))
            z0, z1 = z[0], z[1]
            net_mask = z0 < z1
            rows = torch.nonzero(todo, as_tuple=False).flatten()          # rays left to the sampler
            if rows.numel() > 0:
                s_pts, s_hit, s_z = self.ray_sampler(sdf, cam[rows], dirs[rows], object_mask[rows],
                                                     z0[rows], z1[rows])
                pts[rows] = s_pts
                z0[rows] = s_z
                net_mask[rows] = s_hit
            if not self.training:
                return pts, net_mask, z0
            in_mask = ~net_mask & object_mask & ~todo
            out_mask = ~object_mask & ~todo
            mismatch = in_mask | out_mask
            # rays that never enter the bounding sphere: closest approach to the origin (:896-904)
            left_out = mismatch & ~hit
            z_closest = -(dirs * cam).sum(-1)
            z0 = torch.where(left_out, z_closest, z0)
            pts = torch.where(left_out.unsqueeze(-1), cam + z0.unsqueeze(-1) * dirs, pts)
            sel = torch.nonzero(mismatch & hit, as_tuple=False).flatten()
            if sel.numel() > 0:
                z_min = torch.where(net_mask & out_mask, z0, z_min)
                m_pts, m_z = self.minimal_sdf_points(sdf, cam[sel], dirs[sel], z_min[sel], z_max[sel],
                                                     uniform_steps)
                pts[sel] = m_pts
                z0[sel] = m_z
            return pts, net_mask, z0

    # -- :920-1032 -------------------------------------------------------------------------
    def sphere_tracing(self, sdf, cam, dirs, hit, span):
        """March from the sphere entry forward (end 0) and from the exit backward (end 1) until
        |sdf| <= threshold, the ends cross, or the iteration cap.  cam, dirs (R,3), hit (R,), span
        (R,2).  Returns the end-0 points (R,3), the unfinished end-0 mask (R,), depths (2,R) and
        the initial (min, max) depths."""
        R, dev = dirs.shape[0], dirs.device
        p, st = _lib.ptr, _lib.stream()
        cam, dirs = cam.contiguous(), dirs.contiguous()
        zero = torch.zeros((), device=dev)
        z = torch.where(hit.unsqueeze(0), span.t(), zero).contiguous()              # :933-944
        first_pts = torch.where(hit.unsqueeze(-1), cam + z[0].unsqueeze(-1) * dirs, zero)
        z_min, z_max = z[0].clone(), z[1].clone()                                    # :947-948
        live = hit.unsqueeze(0).expand(2, R).to(torch.uint8).contiguous()
        nxt = torch.zeros((2, R), dtype=torch.float32, device=dev)
        rows = torch.nonzero(hit, as_tuple=False).flatten()
        if rows.numel() > 0:                                                         # :953-959, one call
            both = torch.cat([cam[rows] + z[0][rows].unsqueeze(-1) * dirs[rows],
                              cam[rows] + z[1][rows].unsqueeze(-1) * dirs[rows]])
            nxt[:, rows] = sdf(both).reshape(2, -1)
        cur = torch.empty_like(z)
        slot = torch.empty((2, R), dtype=torch.int32, device=dev)
        todo = torch.empty((2 * R, 3), dtype=torch.float32, device=dev)             # next evaluation's input
        count = torch.zeros((1,), dtype=torch.int32, device=dev)
        no_values = torch.empty((0,), dtype=torch.float32, device=dev)

        def values_of(n):
            return sdf(todo[:n]).reshape(-1).float().contiguous() if n > 0 else no_values

        iters, later = 0, 0
        while True:
            step = 1 if iters < self.sphere_tracing_iters else 0
            _lib.call("iso_raymarch_settle", p(cam), p(dirs), R, p(z), p(cur), p(nxt), p(live),
                      float(self.sdf_threshold), later, step, p(slot), p(todo), p(count), st)
            n = int(count.item())                                                    # :977 (the one host read)
            if not step or n == 0:
                break
            iters += 1
            later = 1
            k = 0
            val = values_of(n)
            _lib.call("iso_raymarch_overshoot", p(cam), p(dirs), R, p(z), p(cur), p(nxt), p(val), 1,
                      1 if k < self.line_step_iters else 0, float(1 - self.line_search_step), p(slot), p(todo),
                      p(count), st)
            while k < self.line_step_iters:                                          # :1004-1025
                n = int(count.item())
                if n == 0:
                    break
                val = values_of(n)
                k += 1
                _lib.call("iso_raymarch_overshoot", p(cam), p(dirs), R, p(z), p(cur), p(nxt), p(val), 0,
                          1 if k < self.line_step_iters else 0,
                          float((1 - self.line_search_step) / (2 ** k)), p(slot), p(todo), p(count), st)
        # every statement that touches the points writes cam + z d (:987-990, :1008-1014); before the
        # first step they are the entry points, zero for the rays that miss the sphere (:927-931)
        pts0 = cam + z[0].unsqueeze(-1) * dirs if iters > 0 else first_pts
        return pts0, live[0].bool(), z, z_min, z_max

    # -- :1034-1112 ------------------------------------------------------------------------
    def ray_sampler(self, sdf, cam, dirs, in_gt, z_lo, z_hi):
        """n_steps uniform samples on [z_lo, z_hi] of the n unfinished rays (already compacted):
        first sample with a negative value -> secant with its predecessor; rays without one (or,
        when training, outside the ground-truth mask) take the sample of lowest value.
        Returns points (n,3), network mask (n,), depths (n,)."""
        n, S = dirs.shape[0], self.n_steps
        lin = torch.linspace(0, 1, steps=S, device=dirs.device).view(1, S)
        zs = z_lo.unsqueeze(-1) + lin * (z_hi - z_lo).unsqueeze(-1)
        P = cam.unsqueeze(1) + zs.unsqueeze(-1) * dirs.unsqueeze(1)
        val = sdf(P.reshape(-1, 3)).reshape(n, S)
        rank = torch.arange(S, 0, -1, device=dirs.device, dtype=torch.float32).view(1, S)
        first_neg = torch.argmin(torch.sign(val) * rank, -1, keepdim=True)          # :1061-1063
        v_first = torch.gather(val, 1, first_neg).squeeze(1)
        in_net = v_first < 0
        # every ray starts from its lowest sample (:1074-1082); the rays that go to the secant
        # below -- the only ones that keep anything of the first-negative sample -- are overwritten
        pick = torch.argmin(val, -1, keepdim=True)
        out_z = torch.gather(zs, 1, pick).squeeze(1)
        out_pts = torch.gather(P, 1, pick.unsqueeze(-1).expand(n, 1, 3)).squeeze(1)
        sec = (in_net & in_gt) if self.training else in_net
        rows = torch.nonzero(sec, as_tuple=False).flatten()
        if rows.numel() > 0:
            k = first_neg[rows]
            km1 = torch.remainder(k - 1, S)                               # index -1 wraps, as in :1096-1099
            z = self.secant(sdf, torch.gather(val[rows], 1, km1).squeeze(1), v_first[rows],
                            torch.gather(zs[rows], 1, km1).squeeze(1), torch.gather(zs[rows], 1, k).squeeze(1),
                            cam[rows], dirs[rows])
            out_pts[rows] = cam[rows] + z.unsqueeze(-1) * dirs[rows]
            out_z[rows] = z
        return out_pts, in_net, out_z

    # -- :1114-1133 ------------------------------------------------------------------------
    def secant(self, sdf, f_lo, f_hi, z_lo, z_hi, cam, dirs):
        """false position on n compacted rays (f_lo > 0 outside, f_hi < 0 inside): one fused
        update (bracket move + next depth + next point) between two network evaluations."""
        p, st = _lib.ptr, _lib.stream()
        n = dirs.shape[0]
        f_lo, f_hi, z_lo, z_hi = [t.float().contiguous().clone() for t in (f_lo, f_hi, z_lo, z_hi)]
        cam, dirs = cam.contiguous(), dirs.contiguous()
        z = torch.empty_like(z_lo)
        mid = torch.empty((n, 3), dtype=torch.float32, device=dirs.device)
        f_mid = None
        for it in range(self.n_secant_steps + 1):
            _lib.call("iso_raymarch_secant", p(cam), p(dirs), n, p(f_lo), p(f_hi), p(z_lo), p(z_hi), p(z),
                      p(f_mid) if f_mid is not None else None, p(mid), st)
            if it < self.n_secant_steps:
                f_mid = sdf(mid).reshape(-1).float().contiguous()
        return z

    # -- :1135-1167 ------------------------------------------------------------------------
    def minimal_sdf_points(self, sdf, cam, dirs, z_min, z_max, uniform_steps=None):
        S = self.n_steps
        if uniform_steps is None:
            uniform_steps = torch.empty(S).uniform_(0.0, 1.0)
        u = uniform_steps.to(device=dirs.device, dtype=torch.float32).view(1, S)
        zs = u * (z_max - z_min).unsqueeze(-1) + z_min.unsqueeze(-1)
        P = cam.unsqueeze(1) + zs.unsqueeze(-1) * dirs.unsqueeze(1)
        j = sdf(P.reshape(-1, 3)).reshape(-1, S).argmin(-1, keepdim=True)
        return torch.gather(P, 1, j.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1), torch.gather(zs, 1, j).squeeze(1)
