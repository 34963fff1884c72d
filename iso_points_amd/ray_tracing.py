"""IDR's two-ended ray tracer: `RayTracing` with the reference's constructor,
call signature and three results (DSS/models/levelset_sampling.py:810-1167).

Every network evaluation goes through the `sdf` callable the caller passes -- build it with
`sdf_models.FusedSdf(model, device)` to run the forward-only fused HIP kernels (SIREN split-fp16
MFMA, IDR x16).  What differs from the reference is the shape of the work handed to that callable:
the reference treats the two marching ends separately and compacts each with boolean masks before
every call (2 network calls, 2 masked scatters and several host syncs per iteration); here both
ends are one (2,R) state, finished rays are held with `where`, and each iteration makes ONE network
call on the still-unfinished rows of the (2R,3) batch (listed by one `nonzero`, whose row count is
also the loop condition the reference reads with `.sum() == 0`).  Per ray the arithmetic is the
same statement for statement and the fused kernels' per-point results do not depend on which
other points share the launch, so the results do not depend on this regrouping.
The interval sampler and the minimal-value search only touch the (few) unfinished / mismatched
rays and keep the reference's compaction.
"""
import torch
import torch.nn as nn

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
        R = dirs.shape[0]
        thr = self.sdf_threshold
        toward = torch.tensor([[1.0], [-1.0]], device=dirs.device)       # end 1 walks backwards
        zero = torch.zeros((), device=dirs.device)

        def evaluate(points, rows, into):
            """network values of the listed rows of the (2R,3) batch, scattered into a copy of `into`"""
            out = into.reshape(-1).clone()
            if rows.numel() > 0:
                out[rows] = sdf(points.reshape(-1, 3)[rows])
            return out.view(2, R)

        def listed(mask):
            return torch.nonzero(mask.reshape(-1), as_tuple=False).flatten()     # (host read: the row count)

        live = hit.unsqueeze(0).expand(2, R).clone()
        z = torch.where(live, span.t(), zero)
        pts = torch.where(live.unsqueeze(-1), cam + z.unsqueeze(-1) * dirs, zero)
        z_min, z_max = z[0].clone(), z[1].clone()
        nxt = evaluate(pts, listed(live), torch.zeros_like(z))
        iters = 0
        while True:
            cur = torch.where(live, nxt, zero)
            cur = torch.where(cur <= thr, zero, cur)
            live = live & (cur > thr)
            rows = listed(live)
            if iters == self.sphere_tracing_iters or rows.numel() == 0:
                break
            iters += 1
            z = z + toward * cur
            pts = cam + z.unsqueeze(-1) * dirs
            nxt = evaluate(pts, rows, torch.zeros_like(z))
            over = nxt < 0                                               # stepped through the surface
            k = 0
            while k < self.line_step_iters:
                rows = listed(over)
                if rows.numel() == 0:
                    break
                back = (1 - self.line_search_step) / (2 ** k)
                z = torch.where(over, z - toward * (back * cur), z)
                pts = torch.where(over.unsqueeze(-1), cam + z.unsqueeze(-1) * dirs, pts)
                nxt = evaluate(pts, rows, nxt)
                over = nxt < 0
                k += 1
            live = live & (z[0] < z[1]).unsqueeze(0)
        return pts[0].clone(), live[0], z.clone(), z_min, z_max

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
        z = -f_lo * (z_hi - z_lo) / (f_hi - f_lo) + z_lo
        for _ in range(self.n_secant_steps):
            f_mid = sdf(cam + z.unsqueeze(-1) * dirs)
            pos, neg = f_mid > 0, f_mid < 0
            z_lo, f_lo = torch.where(pos, z, z_lo), torch.where(pos, f_mid, f_lo)
            z_hi, f_hi = torch.where(neg, z, z_hi), torch.where(neg, f_mid, f_hi)
            z = -f_lo * (z_hi - z_lo) / (f_hi - f_lo) + z_lo
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
