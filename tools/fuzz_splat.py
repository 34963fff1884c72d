"""Fuzz of the splat forward / backward kernels against the oracle (oracle/splat_oracle.py: the C restatement pinned to the
reference's own CPU rasteriser and golden vectors): random splat sets -- sizes from sub-pixel to screen-filling, points
behind the camera and off screen, exact depth ties, duplicates, heavily overfull tiles (the slice + merge path), empty
clouds -- at random image sizes, K, depth thresholds, bin settings.
  forward : idx / zbuf / qvalue / occupancy bit-exact (also the two-stage _rasterize_coarse -> _rasterize_fine)
  backward: gradients within 1e-5 of the oracle's (relative to the largest component), visible set and radius equal
usage: python tools/fuzz_splat.py [n_cases] [first_seed]        exit code 1 on any mismatch (the seed is printed)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
dev = torch.device("cuda:0")


def make_scene(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    N = int(1 + r(1).item() * 3)
    P = int(r(1).item() ** 2 * 30000) + (0 if seed % 11 == 0 else 5)
    pts = r(P, 3) * torch.tensor([2.6, 2.6, 3.0]) - torch.tensor([1.3, 1.3, 0.3])
    if seed % 3 == 0 and P > 50:                            # a dense clump: tiles far over the slice size
        m = P // 2
        pts[:m, :2] = (r(m, 2) - 0.5) * 0.25 + (r(1, 2) - 0.5)
    size = 10.0 ** (-2.5 + 2.0 * r(1).item())               # splat scale of the scene: 0.003 .. 0.3 NDC units
    sx = (r(P) * 0.9 + 0.1) * size
    sy = (r(P) * 0.9 + 0.1) * size
    rho = (r(P) - 0.5) * 1.8
    den = 1 - rho ** 2
    a, c, b = 1 / (sx ** 2 * den), 1 / (sy ** 2 * den), -2 * rho / (sx * sy * den)
    cutoff = r(P) * 1.5 + 0.5
    d = 4 * a * c - b ** 2
    pad = (0.7, 1.0, 1.3)[seed % 3]
    rx, ry = torch.sqrt(4 * c * cutoff / d) * pad, torch.sqrt(4 * a * cutoff / d) * pad
    if P > 20:
        pts[::13, 2] = 1.25                                 # exact depth ties
        k = max(1, P // 50)
        pts[-k:] = pts[:k]                                  # exact duplicates (every field below too)
        for t in (a, b, c, cutoff, rx, ry):
            t[-k:] = t[:k]
    cuts = sorted((r(N - 1) * P).long().tolist()) if N > 1 else []
    first = torch.tensor([0] + cuts, dtype=torch.int64)
    num = torch.tensor(cuts + [P], dtype=torch.int64) - first
    return {"ndc": pts.contiguous(), "ellipse": torch.stack([a, b, c], -1).contiguous(), "cutoff": cutoff.contiguous(),
            "radii": torch.stack([rx, ry], -1).contiguous(), "first": first, "num": num, "N": N, "P": P}


def run(n_cases, seed0):
    """-> number of mismatching cases (each printed with its seed)"""
    from oracle import splat_oracle as SO
    from iso_points_amd.rasterizer import _C, EllipticalRasterizer
    from iso_points_amd.levelset_sampling import with_host_lengths
    failures = 0
    for seed in range(seed0, seed0 + n_cases):
        sc = make_scene(seed)
        g = torch.Generator().manual_seed(seed + 1)
        S = int(16 + torch.rand(1, generator=g).item() * 80)
        K = (1, 3, 5, 8, 8, 16, 33)[seed % 7]
        thres = (0.05, 0.0, 0.5, 10.0)[seed % 4]
        ref = SO.splat_forward(sc["ndc"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first"], sc["num"], thres, S, K)
        dv = {k: sc[k].to(dev) for k in ("ndc", "ellipse", "cutoff", "radii", "first", "num")}
        bad = []
        try:
            got = _C.splat_points(dv["ndc"], dv["ellipse"], dv["cutoff"], dv["radii"], dv["first"], dv["num"], thres, S, K, 0, 0)
            for gt, rf, nm in zip(got, ref, ("idx", "zbuf", "qvalue", "occupancy")):
                if not torch.equal(gt.cpu(), rf):
                    bad.append("%s (%d entries)" % (nm, (gt.cpu() != rf).sum().item()))
            if K <= 32 and seed % 2 == 0:
                binsz = (16, 32)[seed % 4 // 2]
                segs = _C._rasterize_coarse(dv["ndc"], dv["radii"], dv["first"], dv["num"], S, binsz, 1 << 16)
                fine = _C._rasterize_fine(dv["ndc"], dv["ellipse"], dv["cutoff"], dv["radii"], segs, thres, S, binsz, K)
                for gt, rf, nm in zip(fine, ref, ("idx", "zbuf", "qvalue", "occupancy")):
                    if not torch.equal(gt.cpu(), rf):
                        bad.append("fine(coarse) %s (%d entries)" % (nm, (gt.cpu() != rf).sum().item()))
            if sc["P"] > 0 and seed % 2 == 1:
                go = torch.randn(sc["N"], S, S, generator=g)
                go[go.abs() < 1.0] = 0.0
                gz = torch.randn(ref[1].shape, generator=g)
                gz[gz.abs() < 0.5] = 0
                rgrad, vis_ref, rs_ref = SO.splat_backward(sc["ndc"], sc["radii"], ref[0], sc["first"], sc["num"], go, gz, 10.0)
                pts = dv["ndc"].clone().requires_grad_(True)
                first = with_host_lengths(dv["first"], sc["first"].tolist())
                num = with_host_lengths(dv["num"], sc["num"].tolist())
                oi, oz, oq, oo = EllipticalRasterizer.apply(pts, dv["ellipse"], dv["cutoff"], dv["radii"], first, num,
                                                            thres, S, K, 0, 0, 10.0)
                ((oo * go.to(dev)).sum() + (oz * gz.to(dev)).sum()).backward()
                gg = pts.grad.cpu()
                scale = rgrad.abs().max(dim=0).values.clamp(min=1e-30)
                err = ((gg - rgrad).abs().max(dim=0).values / scale)
                if not bool((err < 1e-5).all()) or not bool(torch.isfinite(gg).all()):
                    bad.append("backward rel err %s" % err.tolist())
        except Exception as e:
            bad.append("raised %r" % (e,))
        if bad:
            failures += 1
            print("MISMATCH seed %d (P %d, N %d, S %d, K %d, thres %g): %s" % (seed, sc["P"], sc["N"], S, K, thres, "; ".join(bad)),
                  flush=True)
    print("fuzz_splat: %d cases, %d mismatches" % (n_cases, failures))
    return failures


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1000) else 0)
