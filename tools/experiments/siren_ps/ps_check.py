"""A/B of the point-stationary SIREN step kernel (csrc/siren_ps.hip, ISO_SIREN_PS=1) against k_siren_step_x3:
   python tools/ps_check.py run OUT.pt   -> one evaluation (several list sizes) + a T = 10 projection, saved; timings printed
   python tools/ps_check.py cmp A.pt B.pt -> bit comparison of two such files
The kernel selection is read once per process (environment), hence two processes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))


def run(out):
    from iso_points_amd import _lib
    if os.environ.get("ISO_DEV_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["ISO_DEV_LIB"])
    from iso_points_amd.sdf_models import PackedSiren, Siren
    dev = torch.device("cuda:0")
    res = {}
    for L in (3, 2, 4, 5):
        torch.manual_seed(L)
        m = Siren(hidden_size=256, n_layers=L).to(dev)
        ps = PackedSiren(m, dev)
        sizes = (1, 31, 128, 129, 4097, 100003) if os.environ.get("PS_CHECK_SMALL") else (1, 31, 128, 129, 4097, 100003, 1000000)
        for P in sizes:
            g = torch.Generator().manual_seed(P)
            pts = (torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1) *
                   (0.7 + 0.6 * torch.rand(P, 1, generator=g))).to(dev).contiguous()
            sdf = torch.empty((P,), dtype=torch.float32, device=dev)
            grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
            ws = ps.workspace(P)

            def ev():
                _lib.call("iso_siren_sdf_grad", _lib.ptr(pts), _lib.ptr(sdf), _lib.ptr(grad), P, _lib.ptr(ps.packed), ps.hidden,
                          ps.n_hidden, ps.omega_first, ps.omega_hidden, _lib.ptr(ws), ws.numel(), _lib.stream())
            ev()
            torch.cuda.synchronize()
            res["sdf_L%d_P%d" % (L, P)] = sdf.cpu().clone()
            res["grad_L%d_P%d" % (L, P)] = grad.cpu().clone()
            if P == 1000000:
                ts = []
                for _ in range(7):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    ev()
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
                ts.sort()
                print("L=%d  1M evaluations: %.3f ms (min %.3f)" % (L, ts[len(ts) // 2], ts[0]), flush=True)
        # a projection (device-side lists, moves, compaction)
        P = 120000 if os.environ.get("PS_CHECK_SMALL") else 200000
        g = torch.Generator().manual_seed(7)
        pts = (torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1) * 0.9).to(dev).contiguous()
        outp = torch.empty_like(pts)
        nrm = torch.empty_like(pts)
        mask = torch.empty((P,), dtype=torch.uint8, device=dev)
        ws = ps.workspace(P)
        _lib.call("iso_project_siren", _lib.ptr(pts), _lib.ptr(outp), _lib.ptr(nrm), _lib.ptr(mask), P, _lib.ptr(ps.packed),
                  ps.hidden, ps.n_hidden, ps.omega_first, ps.omega_hidden, 10, 1e-4, _lib.ptr(ws), ws.numel(), _lib.stream())
        torch.cuda.synchronize()
        res["proj_pts_L%d" % L] = outp.cpu().clone()
        res["proj_nrm_L%d" % L] = nrm.cpu().clone()
        res["proj_mask_L%d" % L] = mask.cpu().clone()
    torch.save(res, out)
    print("saved", out, flush=True)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in sorted(A):
        x, y = A[k], B[k]
        same = torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()) if x.is_floating_point() else torch.equal(x, y)
        if not same:
            bad += 1
            d = (x.float() - y.float()).abs()
            print("DIFF %-24s max abs %.3e  (%d of %d elements differ)  max|x| %.3e" % (k, d.max().item(), int((x != y).sum()), x.numel(), x.float().abs().max().item()))
        else:
            print("same %s" % k)
    print("RESULT:", "bit-identical" if bad == 0 else "%d arrays differ" % bad)
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
