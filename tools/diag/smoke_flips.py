"""How many "stop flips" smoke()'s comparison sees, stage by stage (projection alone; resample fed with the oracle's
projection; the chained form smoke() used), for a few cloud sizes / seeds.  usage: python tools/diag/smoke_flips.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths, ProjectionResult  # noqa: E402
from oracle import iso_oracle as O  # noqa: E402

dev = torch.device("cuda:0")


def count(a, b, tol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().amax(-1) / b.abs().max()
    return int((err > tol).sum()), err.numel(), err.max().item()


for threads in (0,):
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    siren = O.fit_siren_to_sphere(O.SirenSDF(hidden_size=128, n_layers=2), steps=200)
    w = torch.cat([p.detach().flatten() for p in siren.parameters()])
    print("threads", threads or "default", "weights checksum %.9e" % w.double().sum().item())
    for P, seed in ((2048, 0), (2048, 1), (2048, 2), (8192, 0)):
        g = torch.Generator().manual_seed(seed)
        pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
        pts = pts + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)
        num = torch.tensor([P])
        ref0 = O.project_points(siren, pts, num, proj_max_iters=10)
        ref = O.resample(siren, ref0.points, ref0.normals, num, sample_iters=1, knn_k=8)
        proj = UniformProjection(knn_k=8)
        gp = pts.to(dev)
        r0 = proj._project_points(siren, gp, full_lengths(gp), proj_max_iters=10)
        res_chain = proj.resample(siren, r0.points, r0.normals, full_lengths(gp), sample_iters=1)
        res_fed = proj.resample(siren, ref0.points.to(dev), ref0.normals.to(dev), full_lengths(gp), sample_iters=1)
        # fixed iteration count (stopping tolerance 1e-30): no stop flips by construction
        f0 = O.project_points(siren, pts, num, proj_max_iters=10, proj_tolerance=1e-30)
        fr = O.resample(siren, f0.points, f0.normals, num, sample_iters=1, knn_k=8, proj_tolerance=1e-30)
        pf = UniformProjection(knn_k=8, proj_tolerance=1e-30)
        g0 = pf._project_points(siren, gp, full_lengths(gp), proj_max_iters=10)
        gr = pf.resample(siren, f0.points.to(dev), f0.normals.to(dev), full_lengths(gp), sample_iters=1)
        grc = pf.resample(siren, g0.points, g0.normals, full_lengths(gp), sample_iters=1)
        torch.cuda.synchronize()
        print("  fixed T: projection %s | resample fed %s | chained %s" % (count(g0.points, f0.points), count(gr.points, fr.points), count(grc.points, fr.points)))
        print("  P=%d seed=%d: projection %s | resample fed with the oracle's projection %s | chained %s"
              % (P, seed, count(r0.points, ref0.points), count(res_fed.points, ref.points), count(res_chain.points, ref.points)))
