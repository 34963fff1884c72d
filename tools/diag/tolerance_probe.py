"""Measured values behind the tolerances of the SIREN parity tests: error quantiles against float64 (ours / the
reference's own float32 golden) and the fraction of stop-flip points of every assert_projection_close call in the
GPU suite.  usage: python tools/diag/tolerance_probe.py"""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_golden import load, siren_from, siren_from_ref
from iso_points_amd import _lib
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from oracle import iso_oracle as O
dev = torch.device("cuda:0")
lib = _lib.load()
for mode in ("split16", "f32"):
    lib.iso_siren_set_gemm_mode(1 if mode == "split16" else 0)
    g = load("proj_siren_fitted.npz")
    m = siren_from(g).to(dev)
    x = g["points"].to(dev)
    r0 = UniformProjection(proj_tolerance=1e-30)._project_points(m, x, full_lengths(x), proj_max_iters=10)
    r64 = O.project_points(copy.deepcopy(m).cpu().double(), g["points"].double(), torch.tensor([g["points"].shape[1]]),
                           proj_max_iters=10, proj_tolerance=1e-30)
    for what, ours, ref, tru in (("normals", r0.normals, g["fixed_normals"], r64.normals), ("points", r0.points, g["fixed_points"], r64.points)):
        scale = tru.abs().max()
        e_ref = ((ref.double() - tru).abs().amax(-1) / scale).view(-1)
        e_our = ((ours.cpu().double() - tru).abs().amax(-1) / scale).view(-1)
        e_g = ((ours.cpu().double() - ref.double()).abs().amax(-1) / ref.abs().max()).view(-1)
        print(mode, "fitted", what, "vs golden max %.2e" % e_g.max().item(),
              " ".join("q%.2f ours %.2e ref %.2e ratio %.2f" % (q, torch.quantile(e_our, q), torch.quantile(e_ref, q),
                                                              torch.quantile(e_our, q) / torch.quantile(e_ref, q)) for q in (0.5, 0.9, 0.99, 1.0)))
    for name in ("siren_ref_128x2.npz", "siren_ref_256x4.npz"):
        g = load(name)
        m = siren_from_ref(g).to(dev)
        x = g["points"].to(dev)
        r = UniformProjection(proj_tolerance=1e-30)._project_points(m, x, full_lengths(x), proj_max_iters=int(g["T"]))
        m64 = copy.deepcopy(siren_from_ref(g)).double()
        r64 = O.project_points(m64, g["points"].double(), torch.tensor([g["points"].shape[1]]), proj_max_iters=int(g["T"]), proj_tolerance=1e-30)
        scale = r64.points.abs().max()
        e_ref = ((g["fixed_points"].double() - r64.points).abs().amax(-1) / scale).view(-1)
        e_our = ((r.points.cpu().double() - r64.points).abs().amax(-1) / scale).view(-1)
        e_g = ((r.points.cpu() - g["fixed_points"]).abs().amax(-1) / g["fixed_points"].abs().max()).view(-1)
        print(mode, name, "frac>1e-5 vs golden %.4f median %.2e" % ((e_g > 1e-5).float().mean().item(), e_g.median().item()),
              " ".join("q%.2f ours %.2e ref %.2e ratio %.2f" % (q, torch.quantile(e_our, q), torch.quantile(e_ref, q),
                                                              torch.quantile(e_our, q) / torch.quantile(e_ref, q)) for q in (0.5, 0.9, 0.99, 1.0)))
lib.iso_siren_set_gemm_mode(1)
