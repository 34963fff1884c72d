"""Per-stage GPU timings of one bench cycle (diagnostic; single GPU)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd.dist import Comm

def main():
    dev = torch.device("cuda:0")
    model = bench.fitted_siren(dev)
    C = bench.Cycle(dev, model, Comm(enabled=False))
    cyc = C.cyc
    def T(name, fn):
        torch.cuda.synchronize(); t = time.time(); r = fn(); torch.cuda.synchronize()
        print("%-28s %9.3f ms" % (name, (time.time()-t)*1e3), flush=True); return r
    for rep in range(2):
        print("--- pass", rep, flush=True)
        r1 = T("project+resample", cyc.project_resample)
        pts, nrm = r1.points[0], r1.normals[0]
        feats = 0.5 * (torch.nn.functional.normalize(nrm, dim=-1) + 1)
        frags, filt = T("splat_forward", lambda: cyc.splat_forward(pts, nrm, feats))
        img = T("composite", lambda: cyc.composite_band(frags, filt))
        alpha = img[..., 3]
        occ_grad = 2.0 * (alpha - cyc.target) / alpha.numel()
        zg = torch.zeros_like(frags.zbuf); zg[..., 0] = 1e-3 / alpha.numel()
        T("backward (vis+median+kernel)", lambda: cyc.backward(frags, filt, occ_grad, zg))
        T("whole step", cyc.step)
main()
