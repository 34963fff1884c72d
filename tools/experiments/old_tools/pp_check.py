import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import iso_oracle as O
from util import cube_cloud
from iso_points_amd import _lib
if os.environ.get("ISO_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
from iso_points_amd.sdf_models import siren_sdf_and_grad
dev = torch.device("cuda:0")
for LL in (1, 2, 3):
    torch.manual_seed(0)
    m = O.SirenSDF(hidden_size=256, n_layers=LL)
    mg = __import__("copy").deepcopy(m).to(dev)
    for NP in (128, 256, 300, 1000, 40000):
        pts = cube_cloud(NP, seed=1)[0]
        sr, gr = O.compute_sdf_and_grad(pts, m)
        out = []
        for rep in range(2):
            s, g = siren_sdf_and_grad(mg, pts.to(dev))
            out.append((s.cpu(), g.cpu()))
        s, g = out[0]
        eg = (g - gr).abs().amax(-1)
        badg = (eg > 1e-4 * gr.abs().max()).nonzero().view(-1)
        bads = ((s - sr).abs() > 1e-5 * sr.abs().max()).nonzero().view(-1)
        same = torch.equal(out[0][1], out[1][1])
        v1 = siren_sdf_and_grad(mg, pts.to(dev), need_grad=False)[0].cpu()
        v2 = siren_sdf_and_grad(mg, pts.to(dev), need_grad=False)[0].cpu()
        print("   value-only: bad %d repeatable=%s" % (int(((v1 - sr).abs() > 1e-5 * sr.abs().max()).sum()), torch.equal(v1, v2)))
        print("L=%d P=%d: bad sdf %d, bad grad %d (first %s, mod128 %s) repeatable=%s" % (
            LL, NP, bads.numel(), badg.numel(), badg[:6].tolist(), sorted(set((badg % 128).tolist()))[:12], same))
