"""RayTracing.forward (levelset_sampling.py:810-1167) on the fused value-only
kernels: one 512 x 512 view (262 144 rays, the image-generation case) and one 2048-pixel training
batch, for the 4x256 SIREN and the 8x512 IDR network; the oracle's masked loop timed on a sample
of the same rays on the host cores.  usage: python tools/raytrace_bench.py [--cpu-sample 4000]"""
import argparse, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools_common import timeit
from oracle import iso_oracle as O   # model definitions + the CPU leg
from util import fitted_siren
from iso_points_amd.ray_tracing import RayTracing
from iso_points_amd.sdf_models import FusedSdf

ap = argparse.ArgumentParser()
ap.add_argument("--cpu-sample", type=int, default=4000)
args = ap.parse_args()
dev = torch.device("cuda:0")


def view(side, seed):
    """pinhole rays through a side x side pixel grid from a camera at distance 3 (fov ~ 40 deg)"""
    c = torch.tensor([[0.0, 0.6, 2.9]])
    u = torch.linspace(-1.1, 1.1, side)
    tgt = torch.stack(torch.meshgrid(u, u, indexing="xy") + (torch.zeros(side, side),), -1).view(1, -1, 3)
    d = torch.nn.functional.normalize(tgt - c.view(1, 1, 3), dim=-1)
    gt = (tgt.view(-1, 3)[:, :2].norm(dim=-1) < 0.72)         # silhouette of a slightly larger sphere
    return c, d, gt


siren = fitted_siren(O, 256, 3, seed=0, fit=200)
torch.manual_seed(0)
idr = O.IdrSDF(hidden_size=512, n_layers=8, skip_in=(4,), num_frequencies=6)
res = {}
for name, net in (("siren4x256", siren), ("idr8x512", idr)):
    cpu_sdf = lambda x, n=net: n.forward(x).sdf.reshape(-1)
    for case, side, training in (("view512_eval", 512, False), ("batch2048_train", 0, True)):
        cam, dirs, gt = view(512, 0)
        if side == 0:
            sel = torch.randperm(dirs.shape[1], generator=torch.Generator().manual_seed(1))[:2048]
            dirs, gt = dirs[:, sel], gt[sel]
        R = dirs.shape[1]
        net_g = net.to(dev)
        fused = FusedSdf(net_g, dev)
        calls = {"n": 0, "pts": 0}

        def counted(x, f=fused, c=calls):
            c["n"] += 1; c["pts"] += x.shape[0]
            return f(x)
        rt = RayTracing().train(training)
        a, b, c = cam.to(dev), gt.to(dev), dirs.to(dev)
        u = torch.rand(100, generator=torch.Generator().manual_seed(2))
        out = rt(sdf=counted, cam_loc=a, object_mask=b, ray_directions=c, uniform_steps=u)
        t = timeit(lambda: rt(sdf=fused, cam_loc=a, object_mask=b, ray_directions=c, uniform_steps=u), warm=1, rep=5)
        x = torch.empty((calls["pts"], 3), device=dev).uniform_(-1, 1)
        t_eval = timeit(lambda: fused(x), warm=1, rep=3)        # the same number of evaluations in one launch
        net.cpu()
        ns = min(args.cpu_sample if name.startswith("siren") else args.cpu_sample // 8, R)
        t0 = time.perf_counter()
        ref = O.ray_tracing(cpu_sdf, cam, gt[:ns], dirs[:, :ns], training=training, uniform_steps=u)
        tc = time.perf_counter() - t0
        agree = (out[1][:ns].cpu() == ref[1]).float().mean().item()
        res["%s_%s" % (name, case)] = {
            "rays": R, "ms": t, "Mrays_s": R / t / 1e3, "sdf_calls": calls["n"], "sdf_points": calls["pts"],
            "same_points_one_launch_ms": t_eval, "hit_fraction": out[1].float().mean().item(),
            "cpu_oracle": {"rays": ns, "s": tc, "Mrays_s": ns / tc / 1e6, "threads": torch.get_num_threads(),
                           "mask_agreement_on_sample": agree}}
        print(name, case, res["%s_%s" % (name, case)], flush=True)
# image look-up (get_tensor_values): 4 masks of 512 x 512, 250 k iso-points each
from iso_points_amd.ray_sampling import get_tensor_values
yy, xx = torch.meshgrid(torch.linspace(-1, 1, 512), torch.linspace(-1, 1, 512), indexing="ij")
img = ((xx * xx + yy * yy) < 0.6).float().view(1, 1, 512, 512).repeat(4, 1, 1, 1).to(dev)
pp = ((torch.rand(4, 250000, 2, generator=torch.Generator().manual_seed(3)) - 0.5) * 2.2).to(dev)
t_ours = timeit(lambda: get_tensor_values(img, pp, squeeze_channel_dim=True), warm=2, rep=10)
t_torch = timeit(lambda: torch.nn.functional.grid_sample(img, pp.unsqueeze(1), mode="bilinear", padding_mode="reflection",
                                                         align_corners=False).squeeze(2).permute(0, 2, 1).squeeze(-1), warm=2, rep=10)
res["get_tensor_values_4x512x512_1M"] = {"ms": t_ours, "torch_grid_sample_same_gpu_ms": t_torch,
                                          "GB_s": (1e6 * 12) / t_ours / 1e6}
print("get_tensor_values", res["get_tensor_values_4x512x512_1M"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "raytrace_bench.json"), "w"), indent=1)
