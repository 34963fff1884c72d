"""Per-stage wall/GPU timings of one bench cycle (diagnostic)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    dev = torch.device("cuda:0")
    t = time.time(); model = bench.fitted_siren(dev); torch.cuda.synchronize(); print("fit %.2fs" % (time.time()-t), flush=True)
    cyc = bench.Cycle(dev, model, P)
    from iso_points_amd.rasterizer import _C, _visible_and_radius, composite
    def T(name, fn):
        torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize()
        print("%-28s %9.3f ms" % (name, (time.time()-t)*1e3), flush=True); return r
    for rep in range(2):
        print("--- pass", rep, flush=True)
        r0 = T("project T10", lambda: cyc._project(cyc.pts0, 10))
        proj = cyc.proj
        def tree():
            proj._create_tree(r0.points, refresh_tree=True, num_points_per_cloud=cyc.num)
        T("create_tree", tree)
        flat = r0.points.reshape(-1, 3)
        diag = (flat.max(dim=0).values - flat.min(dim=0).values).norm()
        inv_sigma = (cyc.num.float() / diag).reshape(1).contiguous()
        moved = T("repulse", lambda: proj.repulsion_step(r0.points, r0.normals, proj._knn_idx, inv_sigma))
        r1 = T("project T3", lambda: cyc._project(moved, 3))
        pts, nrm = r1.points[0], r1.normals[0]
        feats = 0.5 * (torch.nn.functional.normalize(nrm, dim=-1) + 1)
        ss = cyc.splat
        views_c = cyc.views.contiguous()
        flags, off, lens = T("filter", lambda: ss.filter_renderable(pts, nrm, views_c))
        print("   lens", lens, flush=True)
        frags, filt = T("splat.forward (all)", lambda: ss.forward(pts, nrm, cameras=(cyc.views, cyc.projs), features=feats))
        img = T("composite", lambda: composite(frags, filt["scaler"], filt["features"]))
        alpha = img[..., 3]
        occ_grad = 2.0 * (alpha - cyc.target) / alpha.numel()
        print("   occ_grad nonzero frac", (occ_grad != 0).float().mean().item(), "occ mean", alpha.mean().item(), flush=True)
        zbuf_grad = torch.zeros_like(frags.zbuf); zbuf_grad[..., 0] = 1e-3 / alpha.numel()
        vis, rs = T("visible+median", lambda: _visible_and_radius(frags.idx, filt["radii"], filt["first_idx"], filt["num_points"], 10.0))
        print("   rs", rs.tolist(), "visible", vis.sum().item(), flush=True)
        T("backward", lambda: _C._backward(filt["ndc"], filt["radii"], occ_grad, filt["first_idx"], filt["num_points"], visible=vis, rs=rs, idx=frags.idx, grad_zbuf=zbuf_grad))
    if "--cpu" in sys.argv:
        t = time.time(); print(bench.cpu_baseline(model)); print("cpu baseline %.1fs" % (time.time()-t))
main()
