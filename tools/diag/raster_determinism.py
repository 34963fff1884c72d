import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from iso_points_amd.rasterizer import SurfaceSplatting, PointsRasterizationSettings
from oracle import splat_oracle as SO   # camera helpers only
dev = torch.device("cuda:0")
N, S, K, P = 4, 512, 8, 1000000
g = torch.Generator().manual_seed(5)
pts = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
nrm = pts.clone()
views = torch.stack([SO.look_at_view(5.0, 20.0, 90.0 * i) for i in range(N)]).to(dev)
projs = views @ SO.perspective(30.0).to(dev)
ss = SurfaceSplatting(raster_settings=PointsRasterizationSettings(image_size=S, points_per_pixel=K))
ref = None
for r in range(6):
    frags, filt = ss.forward(pts, nrm, cameras=(views, projs))
    torch.cuda.synchronize()
    idx = frags.idx.clone()
    if ref is None:
        ref = idx
        continue
    d = (idx != ref).any(-1)          # (N,S,S)
    n = int(d.sum())
    print("run", r, "pixels that differ:", n)
    if n:
        nz = d.nonzero()
        tiles = set((int(a), int(b) // 16, int(c) // 16) for a, b, c in nz[:2000].tolist())
        print("  tiles:", len(tiles), sorted(tiles)[:8])
        for a, b, c in nz[:12].tolist():
            print("  px", (a, b, c), "lane", ((S - 1 - b) % 16) * 16 + (S - 1 - c) % 16, ref[a, b, c].tolist(), idx[a, b, c].tolist(),
                  ["%.9g" % v for v in frags.zbuf[a, b, c].tolist()])
