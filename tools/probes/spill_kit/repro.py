"""usage: python repro.py NAME   (NAME = default | fold | haz | ...: which library; `default` writes the reference)"""
import sys, os, torch
ROOT = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from iso_points_amd import _lib
name = sys.argv[1]
if name != "default":
    _lib.LIB_PATH = os.path.join(ROOT, "libiso_%s.so" % name)
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
REF = "/tmp/spill_ref_idx.pt"
ref = torch.load(REF).to(dev) if name != "default" else None
runs = []
for r in range(8):
    frags, filt = ss.forward(pts, nrm, cameras=(views, projs))
    torch.cuda.synchronize()
    idx, zb, qv = frags.idx.clone(), frags.zbuf.clone(), frags.qvalue.clone()
    if ref is None:
        ref = idx
        torch.save(idx.cpu(), REF)
        torch.save(qv.cpu(), REF + ".q")
    if name != "default" and r == 0:
        refq = torch.load(REF + ".q").to(dev)
    d = (idx != ref).any(-1)
    runs.append(d)
    n = int(d.sum())
    msg = "%s run %d: pixels that differ from the reference: %d" % (name, r, n)
    if n:
        nz = d.nonzero()
        a = idx[d]; b = ref[d]; z = zb[d]
        first = (a != b).float().argmax(-1)                         # first list position that differs
        tie = (z[:, 1:] == z[:, :-1]) & (z[:, 1:] >= 0)
        has_tie = tie.any(-1)
        dup = (a[:, 1:] == a[:, :-1]) & (a[:, 1:] >= 0)
        # is the first differing position the first entry of a tie pair?
        at_tie = torch.gather(torch.cat([tie, torch.zeros_like(tie[:, :1])], 1), 1, first[:, None])[:, 0]
        qa, qb = qv[d], refq[d]
        qj = torch.gather(qa, 1, first[:, None])[:, 0]                       # q at the first wrong position
        qref = torch.gather(qb, 1, first[:, None])[:, 0]
        qnext = torch.gather(qa, 1, (first + 1).clamp(max=K - 1)[:, None])[:, 0]
        msg += "; q at the wrong slot: = the reference's (the missing point's) %d, = the next slot's (the duplicate's) %d, neither %d" % (
            int((qj == qref).sum()), int(((qj == qnext) & (qj != qref)).sum()), int(((qj != qref) & (qj != qnext)).sum()))
        msg += "; with a depth tie %d, first difference AT a tie's first entry %d, duplicate id in the list %d; first-difference positions %s" % (
            int(has_tie.sum()), int(at_tie.sum()), int(dup.any(-1).sum()), torch.bincount(first, minlength=K).tolist())
        if r == 0 or r == 7:
            for k in range(min(4, n)):
                msg += "\n   px %s got %s want %s z %s" % (nz[k].tolist(), a[k].tolist(), b[k].tolist(), ["%.9g" % v for v in z[k].tolist()])
    print(msg)
    if "post" in name:
        import ctypes, struct
        L = ctypes.CDLL(_lib.LIB_PATH)
        out = (ctypes.c_uint * 16)(); L.iso_dbg_counts(out)
        dump = (ctypes.c_int * 4096)(); L.iso_dbg_dump(dump)
        print("   post-merge check: %d pixels whose register merge differs from the all-from-scratch merge" % out[0])
        f = lambda i: struct.unpack("f", struct.pack("i", i))[0]
        for k in range(min(out[0], 6) if r in (0, 7) else 0):
            o = dump[k * 64:(k + 1) * 64]
            print("   slice %d of %d lane %d tile %d\n      register merge %s\n      scratch merge  %s\n      z (registers)  %s\n      own list (scratch) ids %s\n      own list (scratch) z   %s" % (
                o[0], o[1], o[2], o[3], list(o[8:16]), list(o[16:24]), ["%.9g" % f(x) for x in o[24:32]], list(o[32:40]), ["%.9g" % f(x) for x in o[40:48]]))
    if "dbg" in name:
        import ctypes
        L = ctypes.CDLL(_lib.LIB_PATH)
        out = (ctypes.c_uint * 16)()
        L.iso_dbg_counts(out)
        print("   dbg: own list: z order %d, tie order %d, duplicates %d | other lists: tie order %d, duplicates %d | merges %d | merged list: tie order %d, duplicates %d" % (
            out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7]))
same = all(bool((runs[0] == x).all()) for x in runs[1:])
print("%s: the set of differing pixels is %s over the 8 runs" % (name, "THE SAME" if same else "DIFFERENT"))
