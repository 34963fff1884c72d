"""Ray queries: value-only vs value+gradient evaluation rate of the fused SDF
kernels, and SphereTracing.project_points on 1 M rays (4 views of 512 x 512) for the analytic
sphere, the 4x256 SIREN and the 8x512 IDR network; the oracle's loop timed on a sample of the same
rays on the host cores.  usage: python tools/trace_bench.py [--cpu-sample 20000]"""
import argparse, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_common import timeit
from oracle import iso_oracle as O   # model definitions + the CPU leg
from util import fitted_siren
from iso_points_amd.levelset_sampling import SphereTracing
from iso_points_amd.sdf_models import SphereSDF, FusedSdf, siren_sdf_and_grad, idr_sdf_and_grad

ap = argparse.ArgumentParser()
ap.add_argument("--cpu-sample", type=int, default=20000)
ap.add_argument("--rays", type=int, default=4 * 512 * 512)
args = ap.parse_args()
dev = torch.device("cuda:0")
res = {}


def rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    views = torch.nn.functional.normalize(torch.tensor([[0.0, 0.3, 1.0], [1.0, 0.3, 0.0], [0.0, 0.3, -1.0], [-1.0, 0.3, 0.0]]), dim=-1) * 3
    cam = views[torch.arange(n) % 4]
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.8
    d = torch.nn.functional.normalize(tgt - cam, dim=-1)
    b = (d * cam).sum(-1)
    disc = (b * b - ((cam * cam).sum(-1) - 1.05 ** 2)).clamp_min(0)
    return cam + (-b - disc.sqrt())[:, None] * d, d


r0, d = rays(args.rays, 1)
r0g, dg = r0.to(dev), d.to(dev)
siren = fitted_siren(O, 256, 3, seed=0, fit=200)
torch.manual_seed(0)
idr = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6)
models = [("sphere", O.SphereSDF(radius=0.6), SphereSDF(radius=0.6).to(dev), args.rays),
          ("siren4x256", siren, None, args.rays), ("idr8x512", idr, None, args.rays // 4)]
# evaluation rates
pts = r0g[:1000000].contiguous()
for name, fn, m, flop in (("siren4x256", siren_sdf_and_grad, siren, 0.79e6), ("idr8x512", idr_sdf_and_grad, idr, 7.3e6)):
    mg = m.to(dev)
    P = pts.shape[0] if name.startswith("siren") else 300000
    x = pts[:P]
    tf = timeit(lambda: fn(mg, x), warm=1, rep=5)
    tv = timeit(lambda: fn(mg, x, need_grad=False), warm=1, rep=5)
    res["eval_" + name] = {"points": P, "value_grad_ms": tf, "value_only_ms": tv,
                           "value_grad_Mevals_s": P / tf / 1e3, "value_only_Mevals_s": P / tv / 1e3,
                           "value_grad_TFLOPs": P * flop / tf / 1e9, "value_only_TFLOPs": P * flop / 2 / tv / 1e9}
    print(name, res["eval_" + name], flush=True)
for name, m_cpu, m_gpu, n in models:
    m_gpu = m_gpu if m_gpu is not None else m_cpu.to(dev)
    st = SphereTracing(proj_max_iters=10)
    a, b = r0g[:n].contiguous(), dg[:n].contiguous()
    out = st.project_points(a, b, m_gpu)
    t = timeit(lambda: st.project_points(a, b, m_gpu), warm=1, rep=5)
    ns = min(args.cpu_sample, n) if name != "idr8x512" else min(args.cpu_sample // 10, n)
    m_c = m_cpu.cpu() if hasattr(m_cpu, "cpu") else m_cpu
    t0 = time.perf_counter()
    ref = O.sphere_trace(m_c, r0[:ns], d[:ns], proj_max_iters=10)
    tc = time.perf_counter() - t0
    agree = (out["mask"][:ns].cpu() == ref["mask"]).float().mean().item()
    res["trace_" + name] = {"rays": n, "ms": t, "Mrays_s": n / t / 1e3, "hit_fraction": out["mask"].float().mean().item(),
                            "cpu_oracle_rays": ns, "cpu_oracle_s": tc, "cpu_oracle_Mrays_s": ns / tc / 1e6,
                            "cpu_threads": torch.get_num_threads(), "mask_agreement_on_sample": agree}
    print(name, res["trace_" + name], flush=True)
# ray -> nearest point (combined_modeling.py:336-352): fused search vs the dense (R,M) statement on the GPU
from iso_points_amd.ray_sampling import ray_nearest_point
cam = torch.tensor([0.0, 0.9, 2.8])
for R, M in ((4096, 100000), (16384, 1000000)):
    rr = torch.nn.functional.normalize((torch.rand(R, 3) - 0.5) * 1.6 - cam, dim=-1).to(dev)
    pp = torch.nn.functional.normalize(torch.randn(M, 3), dim=-1).to(dev)
    t = timeit(lambda: ray_nearest_point(rr, cam, pp), warm=1, rep=5)

    def dense():
        out = []
        for ch in torch.split(rr, 2048):                       # (2048, M) f32 chunks: 8 GB at M = 1e6
            pC = pp - cam.to(dev).view(1, 3)
            sq = (pC[None] * ch[:, None]).sum(-1) ** 2
            dd = (pC ** 2).sum(-1).unsqueeze(0) - sq
            out.append(torch.gather(sq, 1, torch.topk(dd, k=1, dim=1, largest=False)[1]))
        return torch.cat(out)
    td = timeit(dense, warm=1, rep=3)
    res["ray_nearest_%dx%d" % (R, M)] = {"fused_ms": t, "Gpairs_s": R * M / t / 1e6, "torch_dense_on_gpu_ms": td}
    print("ray_nearest", R, M, res["ray_nearest_%dx%d" % (R, M)], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "trace_bench.json"), "w"), indent=1)
