"""bench.py -- iso-point cycle throughput on MI355X.

Metric (BASELINE.json): Mpoints/s of the full iso-point cycle (project + resample + splat),
1M points; workload = configs[2] "1M points project+resample + DSS EWA splat fwd/bwd at
512x512x4 views", SDF = the 4-layer x 256 SIREN of configs[1] (SURVEY 8(d) cfg 3b).

One step = one pass of the hot path over the same synthetic batch:
  _project_points(T=10) -> resample(sample_iters=1) [= FRNN K+1=9, one tangent-plane repulsion,
  _project_points(T=3)] -> filter / K=7 FRNN / per-point EWA set-up for 4 views -> splat forward
  512^2 x 4 (K=8) -> compositing -> backward (occupancy + zbuf gradients).
Inputs are resident in HBM before the timed region; every step restarts from the same
initial cloud so the work per step is constant.

python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run,
one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P_TOTAL = 1000000
IMAGE = 512
VIEWS = 4
KPIX = 8
HIDDEN, LAYERS = 256, 3
FLOP_PER_EVAL = 2.0 * (2 * LAYERS * HIDDEN * HIDDEN + 2 * 3 * HIDDEN + 2 * HIDDEN)   # 0.79 MFLOP
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense f32-input MFMA
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: dense bf16 MFMA (2.5 PF); 16x the f32 rate
SUSTAINED_FP16_TFLOPS = 2516.6 * 1.66 / 2.4   # measured: dense fp16 MFMA on all 256 CUs clocks at 1.66 GHz (tools/probes/clock.hip)
SUSTAINED_FP16_RANDOM_TFLOPS = 1580.0         # the same with operands that change with every MFMA and are random fp16 numbers
                                              # (tools/probes/mfma_power.hip, profiles/r04_mfma_power_probe.txt, r05_power_siren.txt: 1.50-1.55 GHz)
X3_PASSES = 3                      # fp16 MFMA passes per f32 product in the split-operand mode (siren_x3.hip): both
                                   # operands cut into two fp16 numbers, W_l x_h + W_h x_l + W_h x_h; the fp16 and bf16
                                   # MFMA peaks are equal
PEAK_HBM_TBS = 8.0


def sphere_cloud(P, seed, device):
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    return (p + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)).to(device)


def fitted_siren(device, steps=300, seed=0):
    """Siren(3->256x4->1) fitted to the unit-sphere SDF (SURVEY 8(d) cfg 2: 'weights fitted to the
    sphere for convergence realism'); untimed set-up, deterministic."""
    from iso_points_amd.sdf_models import Siren
    torch.manual_seed(seed)
    m = Siren(dim=3, hidden_size=HIDDEN, n_layers=LAYERS, c_dim=0).to(device)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for _ in range(steps):
        x = ((torch.rand(4096, 3, generator=g) - 0.5) * 3.0).to(device)
        y = x.norm(dim=-1, keepdim=True) - 1.0
        loss = ((m(x).sdf - y) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    for p in m.parameters():
        p.requires_grad_(False)
    return m.eval()


def cameras(device):
    from iso_points_amd.cameras import look_at_view, perspective
    views = torch.stack([look_at_view(3.0, 20.0, 90.0 * i) for i in range(VIEWS)]).to(device)
    projs = views @ perspective(30.0).to(device)
    return views, projs


class Cycle(object):
    """The hot path = iso_points_amd.dist.IsoCycle (the same class for 1 and N GPUs), plus HIP
    events around the SIREN projections for the roofline figure."""

    def __init__(self, device, model, comm):
        from iso_points_amd.dist import IsoCycle, sphere_silhouette
        from iso_points_amd.rasterizer import PointsRasterizationSettings
        self.dev, self.model, self.comm = device, model, comm
        rs = PointsRasterizationSettings(image_size=IMAGE, points_per_pixel=KPIX, cutoff_threshold=1.0,
                                         depth_merging_threshold=0.05, radii_backward_scaler=10,
                                         backface_culling=True, Vrk_isotropic=True, bin_size=None)
        views, projs = cameras(device)
        from iso_points_amd.dist import slab_order
        pts0 = sphere_cloud(P_TOTAL, seed=0, device=device)          # identical on every rank
        # the job's point order: x-slab major, z-order curve inside a slab (one sort at set-up; ISO_BENCH_ORDER=x
        # keeps the generator's order inside a slab, for A/B measurements)
        local = None if os.environ.get("ISO_BENCH_ORDER", "cell") == "x" else "cell"
        pts0 = pts0[:, slab_order(pts0[0], comm.world, local=local)].contiguous()
        self.cyc = IsoCycle(model, pts0, views, projs, raster_settings=rs, knn_k=8, comm=comm,
                            target=sphere_silhouette(IMAGE, VIEWS, 3.0, 30.0, device))
        if comm.world > 1:
            self.cyc.calibrate()                                     # untimed: sizes of the exchange buffers
        self.cyc.marks = True
        self.cyc.use_graphs = os.environ.get("ISO_BENCH_GRAPHS", "1") != "0"
        comm.on_mark = self._on_mark
        self.ev = []
        self.timed = False
        self._open = None
        self._count_hook = None

    def _on_mark(self, name, T):
        """HIP events on the launch stream around the two Newton projections of a cycle."""
        if self._count_hook is not None:
            if name == "project_end":
                self._count_hook(T)
            return
        if not self.timed or name not in ("project_begin", "project_end"):
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if name == "project_begin":
            self._open = e
        else:
            self.ev.append((self._open, e, T))

    def step(self):
        return self.cyc.step()

    def siren_stats(self):
        """(ms, launches) of the timed SIREN projections: HIP events on the launch stream."""
        from iso_points_amd import _lib
        lib = _lib.load()
        ms = sum(a.elapsed_time(b) for a, b, _ in self.ev)
        # launches really issued: the late Newton iterations of a projection are ONE tail launch (iso_siren_set_tail_from)
        launches = sum(lib.iso_siren_step_launches(HIDDEN, LAYERS, T) if lib.iso_siren_get_gemm_mode() == 1 else T + 1
                       for _, _, T in self.ev)
        return ms, launches

    def active_counts(self):
        """Point-evaluations of one cycle on this rank, from the kernel's own device-side
        active-list counters (one extra untimed, eager pass of the two projections)."""
        from iso_points_amd import _lib
        lib = _lib.load()
        cyc = self.cyc
        n = cyc.pts0_local.shape[1]
        off = lib.iso_project_siren_counts_offset(n, HIDDEN, LAYERS)
        counts = []

        def hook(T):
            torch.cuda.synchronize()
            c = cyc.proj._packed_cache._ws[off:off + 64 * 4].view(torch.int32).tolist()
            counts.append([n] + c[1:T + 1])
        self._count_hook = hook
        cyc.run(cyc.project_resample())
        self._count_hook = None
        return counts


def measured_traffic(x3):
    """HBM-side bytes per launch of the dominant kernel.  bench.py cannot run rocprofv3 on itself: the figure comes
    from the PMC passes of THIS command in an earlier builder run (tools/pmc_run.sh -> tools/traffic_from_pmc.py ->
    profiles/rNN_traffic.json; 2*FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction).  Returns (bytes, source)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            return d["k_siren_step_x3" if x3 else "k_siren_step"]["bytes_per_launch"], "profiles/" + os.path.basename(f)
        except Exception:
            continue
    return None, None


def checked(cyc, comm, dev):
    """One more, untimed cycle whose device-side overflow flags (pair list, packed rows, halo export / import,
    uncertified halo queries) are read back: a truncated raster or neighbour list must not produce a number.
    Raises on every rank if any rank overflowed; returns the usage dict."""
    out = cyc.cyc.step()
    err, u = None, None
    try:
        u = cyc.cyc.check(out[4])
    except RuntimeError as e:
        err = str(e)
    bad = comm.max_int(1 if err else 0, dev)
    if bad:
        raise RuntimeError("bench.py: capacity overflow inside the timed cycle -- the figure would be invalid: %s" % err)
    return u


def f32_mode_cycle(dev, model, comm, steps=3):
    """The same cycle with the hidden-layer products on the f32 matrix cores (iso_siren_set_gemm_mode(0)):
    printed beside the split-fp16 headline."""
    from iso_points_amd import _lib
    lib = _lib.load()
    lib.iso_siren_set_gemm_mode(0)
    old_graphs = os.environ.get("ISO_BENCH_GRAPHS")
    try:
        os.environ["ISO_BENCH_GRAPHS"] = "0"
        cyc = Cycle(dev, model, comm)
        cyc.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cyc.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        checked(cyc, comm, dev)
    finally:
        lib.iso_siren_set_gemm_mode(1)
        if old_graphs is None:
            os.environ.pop("ISO_BENCH_GRAPHS", None)
        else:
            os.environ["ISO_BENCH_GRAPHS"] = old_graphs
    return {"ms_per_step": round(ms, 4), "value": round(P_TOTAL / (ms * 1e-3) / 1e6, 3), "unit": "Mpoints/s",
            "note": "hidden-layer products on v_mfma_f32_16x16x4_f32 (k_siren_step<16>), everything else unchanged"}


def generator_order_cycle(dev, model, comm, steps=3):
    """The headline cycle on the cloud in the order the generator emitted it (no x-slab / z-order sort at set-up):
    what a caller pays who hands over an unordered cloud and does not re-sort."""
    old = os.environ.get("ISO_BENCH_ORDER")
    os.environ["ISO_BENCH_ORDER"] = "x"
    try:
        cyc = Cycle(dev, model, comm)
        for _ in range(2):
            cyc.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cyc.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        checked(cyc, comm, dev)
    finally:
        if old is None:
            os.environ.pop("ISO_BENCH_ORDER", None)
        else:
            os.environ["ISO_BENCH_ORDER"] = old
    return {"ms_per_step": round(ms, 4), "value": round(P_TOTAL / (ms * 1e-3) / 1e6, 3), "unit": "Mpoints/s",
            "note": "same cycle, input cloud in generator (random) order"}


OPAPI_STEP_HOOK = None


def operator_api_cycle(dev, model, steps=3):
    """The same configs[2] workload through the reference's OPERATOR signatures, the way train_mvr.py would drive them
    -- no IsoCycle, no fused orchestration, no graphs, host reads where the reference's API has them:
    UniformProjection.project_points(skip_upsampling=True) [levelset_sampling.py:353-439: project T=10, keep the
    converged points, resample (tree + repulsion + project T=3)] -> SurfaceSplatting.forward on points that require grad
    [rasterizer.py:584-661] -> composite [renderer.py:36-82] -> the cycle's loss -> autograd .backward() down to the
    world points.  (project_points drops the points that did not converge before it resamples, as the reference does;
    the headline cycle resamples all of them.)"""
    from iso_points_amd.levelset_sampling import UniformProjection, mask_padded_to_list
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting, composite
    from iso_points_amd.dist import sphere_silhouette, slab_order
    rs = PointsRasterizationSettings(image_size=IMAGE, points_per_pixel=KPIX, cutoff_threshold=1.0,
                                     depth_merging_threshold=0.05, radii_backward_scaler=10, backface_culling=True,
                                     Vrk_isotropic=True, bin_size=None)
    views, projs = cameras(dev)
    pts0 = sphere_cloud(P_TOTAL, seed=0, device=dev)
    pts0 = pts0[:, slab_order(pts0[0], 1, local="cell")].contiguous()       # the headline's input order
    target = sphere_silhouette(IMAGE, VIEWS, 3.0, 30.0, dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
    ss = SurfaceSplatting(cameras=(views, projs), raster_settings=rs)
    kept = [0]

    def step():
        out = proj.project_points(pts0, model, skip_upsampling=True)
        # the converged points, as combined_modeling.py:452-453 takes them (mask_padded_to_list of DSS/utils)
        x = mask_padded_to_list(out["levelset_points"], out["mask"])[0].detach().requires_grad_(True)
        nrm = mask_padded_to_list(out["levelset_normals"], out["mask"])[0]
        kept[0] = x.shape[0]
        frags, filt = ss.forward(x, nrm)
        feat = 0.5 * (torch.nn.functional.normalize(filt["normals"], dim=-1) + 1.0)
        img = composite(frags, filt["scaler"], feat)
        loss = ((img[..., 3] - target) ** 2).mean() + 1e-3 * frags.zbuf[..., 0].mean()
        loss.backward()
        return x.grad

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # median of the steps, each timed on its own: an eager path allocates its workspaces per call, and one step that
    # misses the caching allocator (seen once in five bench runs: 23 ms instead of 16) would skew a mean of three
    times = []
    for _ in range(max(steps, 5)):
        if OPAPI_STEP_HOOK is not None:
            OPAPI_STEP_HOOK()                    # tools/opapi_only.py trace: a marker kernel between steps
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    ms = times[len(times) // 2]
    return {"ms_per_step": round(ms, 4), "value": round(P_TOTAL / (ms * 1e-3) / 1e6, 3), "unit": "Mpoints/s",
            "ms_per_step_min_max": [round(times[0], 4), round(times[-1], 4)],
            "points_kept_after_projection": kept[0],
            "note": "UniformProjection.project_points(skip_upsampling=True) -> SurfaceSplatting.forward -> composite -> "
                    "loss.backward(): the reference's operator signatures, eager, with their host reads; no IsoCycle, no graphs"}


def operator_api_small(dev, model, P=24000, steps=30):
    """The reference's REAL working set through the operator signatures: CombinedModel keeps 5 000 -> 24 000 iso-points
    and renders one view per step (combined_modeling.py:75,82).  24 000 points, 1 view, 512^2, K = 8:
    project_points(skip_upsampling=True) -> SurfaceSplatting.forward -> composite -> loss.backward().  At this size a
    step is bound by launch count and by the host reads the reference's return types need, so the line carries both:
    kernels per step and device -> host copies per step (one step under torch.profiler), next to the median step time."""
    from iso_points_amd.levelset_sampling import UniformProjection, mask_padded_to_list
    from iso_points_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting, composite
    from iso_points_amd.cameras import look_at_view, perspective
    from iso_points_amd.dist import sphere_silhouette
    rs = PointsRasterizationSettings(image_size=IMAGE, points_per_pixel=KPIX, cutoff_threshold=1.0,
                                     depth_merging_threshold=0.05, radii_backward_scaler=10, backface_culling=True,
                                     Vrk_isotropic=True, bin_size=None)
    views = torch.stack([look_at_view(3.0, 20.0, 0.0)]).to(dev)
    projs = views @ perspective(30.0).to(dev)
    pts0 = sphere_cloud(P, seed=0, device=dev)
    target = sphere_silhouette(IMAGE, 1, 3.0, 30.0, dev)
    proj = UniformProjection(proj_max_iters=10, proj_tolerance=5e-5, knn_k=8, sample_iters=1)
    ss = SurfaceSplatting(cameras=(views, projs), raster_settings=rs)

    def step():
        out = proj.project_points(pts0, model, skip_upsampling=True)
        x = mask_padded_to_list(out["levelset_points"], out["mask"])[0].detach().requires_grad_(True)
        nrm = mask_padded_to_list(out["levelset_normals"], out["mask"])[0]
        frags, filt = ss.forward(x, nrm)
        feat = 0.5 * (torch.nn.functional.normalize(filt["normals"], dim=-1) + 1.0)
        img = composite(frags, filt["scaler"], feat)
        loss = ((img[..., 3] - target) ** 2).mean() + 1e-3 * frags.zbuf[..., 0].mean()
        loss.backward()
        return x.grad

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    ms = times[len(times) // 2]
    kernels = d2h = gpu_us = None
    try:                                             # one more step under the profiler: what was launched, what was read
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        copies = [e for e in evs if "memcpy" in e.name.lower()]
        d2h = sum(1 for e in copies if "dtoh" in e.name.lower() or "device -> host" in e.name.lower()
                  or "devicetohost" in e.name.lower())
        kernels = len(evs) - len(copies) - sum(1 for e in evs if "memset" in e.name.lower())
        gpu_us = round(sum(e.device_time if hasattr(e, "device_time") else e.cuda_time for e in evs), 1)
    except Exception as exc:                         # the line still carries the time
        kernels = "profiler unavailable: %s" % type(exc).__name__
    return {"ms_per_step": round(ms, 4), "ms_per_step_min_max": [round(times[0], 4), round(times[-1], 4)],
            "points": P, "views": 1, "image": IMAGE, "points_per_pixel": KPIX,
            "kernels_per_step": kernels, "host_reads_per_step": d2h, "gpu_busy_us_per_step": gpu_us,
            "note": "the reference's working set (combined_modeling.py:75,82): 24 000 iso-points, one view per step; "
                    "eager operator API with its host reads; kernels and device->host copies of ONE step counted by "
                    "torch.profiler"}


def analytic_cycle(dev, comm, args):
    """SURVEY 8(d) cfg 3a: the same cycle with the analytic sphere SDF -- the HBM-bound variant.
    Reported beside the headline (cfg 3b), not instead of it."""
    from iso_points_amd.sdf_models import SphereSDF
    cyc = Cycle(dev, SphereSDF().to(dev), comm)
    cyc.cyc.marks = False         # the marks only bracket the SDF kernel for the headline's roofline figure; each one is a
                                  # graph-segment boundary (~9 us of idle GPU), and this cycle has no SDF kernel to bracket
    for _ in range(max(args.warmup, 1)):
        cyc.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cyc.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    checked(cyc, comm, dev)
    gb = 1.19                                            # algorithmic bytes of one cfg-3a cycle
    return {"ms_per_step": round(ms, 4), "value": round(P_TOTAL / (ms * 1e-3) / 1e6, 3), "unit": "Mpoints/s",
            "roofline": {"bound": "hbm", "achieved": round(gb / (ms * 1e-3) / 1e3, 4), "peak": PEAK_HBM_TBS,
                         "unit": "TB/s", "frac": round(gb / (ms * 1e-3) / 1e3 / PEAK_HBM_TBS, 4),
                         "note": "1.19 GB algorithmic bytes per cycle (SURVEY 8(d)); every kernel above 0.1 ms of "
                                 "this cycle is instruction-bound, not HBM-bound (DESIGN.md 4.1)"}}


def cpu_baseline(gpu_model):
    """The oracle (CPU restatement of the reference's pure-PyTorch path + C rasteriser) timed on
    the host cores of this box: (1) the whole cycle on a bounded sample of the same
    workload (same fitted SIREN weights) = `value`; (2) the point stages one by one at FULL size with
    the analytic sphere SDF (project T=10, FRNN K+1=9, one repulsion step, project T=3) -- the FRNN of
    1 M points by scipy's cKDTree (the oracle's exact search is a chunked brute force, O(P^2));
    (3) the SIREN projection on 100 k points (its cost is linear in P)."""
    from oracle import iso_oracle as O
    from oracle import splat_oracle as SO
    ncores = torch.get_num_threads()
    P, S, V = 50000, 256, 1
    m = O.SirenSDF(hidden_size=HIDDEN, n_layers=LAYERS)
    with torch.no_grad():
        src = [l.linear for l in list(gpu_model.net)[:-1]] + [gpu_model.net[-1]]
        for dst, s_ in zip(m.lins, src):
            dst.weight.copy_(s_.weight.detach().cpu())
            dst.bias.copy_(s_.bias.detach().cpu())
    g = torch.Generator().manual_seed(0)
    pts = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    pts = pts + 0.05 * (torch.rand(1, P, 3, generator=g) - 0.5)
    num = torch.tensor([P])
    t0 = time.perf_counter()
    r0 = O.project_points(m, pts, num, proj_max_iters=10)
    r1 = O.resample(m, r0.points, r0.normals, num, sample_iters=1, knn_k=8)
    t_pr = time.perf_counter() - t0
    p, n = r1.points[0], r1.normals[0]
    view = SO.look_at_view(3.0, 20.0, 0.0)
    M44 = view @ SO.perspective(30.0)
    keep = SO.filter_renderable(p, n, view)
    pf, nf = p[keep], n[keep]
    h = SO.vrk_h(pf[None], torch.tensor([pf.shape[0]]), 0.2)
    info = SO.per_point_info(pf, nf, h, M44, S)
    ndc = SO.transform_to_ndc(pf, view, M44)
    first, numv = torch.tensor([0]), torch.tensor([pf.shape[0]])
    ts = time.perf_counter()
    idx, zb, qv, occ = SO.splat_forward(ndc, info["ellipse_params"], info["cutoff_threshold"], info["radii"],
                                        first, numv, 0.05, S, KPIX)
    t_fwd = time.perf_counter() - ts
    fr = SO.PointFragments(idx, zb, qv, SO.gather_scaler(info["scaler"], idx), occ)
    img = SO.composite(fr, 0.5 * (torch.nn.functional.normalize(nf, dim=-1) + 1))
    go = 2.0 * (img[..., 3] - 0.5) / img[..., 3].numel()
    gz = torch.zeros_like(zb)
    ts = time.perf_counter()
    SO.splat_backward(ndc, info["radii"], idx, first, numv, go, gz, 10.0)
    t_bwd = time.perf_counter() - ts
    t_all = time.perf_counter() - t0
    stages = {"sample_cycle_s": round(t_all, 2), "sample_project_resample_s": round(t_pr, 2),
              "sample_splat_forward_s": round(t_fwd, 2), "sample_splat_backward_s": round(t_bwd, 2),
              "sample": "%d points, splat %dx%dx%d view" % (P, S, S, V)}
    # (2) full-size point stages, analytic SDF
    try:
        import numpy as np
        from scipy.spatial import cKDTree
        PF = P_TOTAL
        g = torch.Generator().manual_seed(0)
        full = torch.nn.functional.normalize(torch.randn(1, PF, 3, generator=g), dim=-1)
        full = full + 0.05 * (torch.rand(1, PF, 3, generator=g) - 0.5)
        numf = torch.tensor([PF])
        sph = O.SphereSDF()
        ts = time.perf_counter()
        q0 = O.project_points(sph, full, numf, proj_max_iters=10)
        stages["full_project_T10_sphere_s"] = round(time.perf_counter() - ts, 2)
        r = float(O.search_radius(q0.points, numf, 8))
        ts = time.perf_counter()
        tree = cKDTree(q0.points[0].numpy())
        dd, ii = tree.query(q0.points[0].numpy(), k=9, distance_upper_bound=r, workers=-1)
        stages["full_frnn_K9_ckdtree_s"] = round(time.perf_counter() - ts, 2)
        ii = torch.from_numpy(np.where(np.isfinite(dd), ii, -1).astype(np.int64))[None, :, 1:]
        diag = (q0.points.view(-1, 3).max(0).values - q0.points.view(-1, 3).min(0).values).norm().item()
        ts = time.perf_counter()
        moved = O.repulsion_step(q0.points, torch.nn.functional.normalize(q0.normals, dim=-1), ii, numf / diag)
        stages["full_repulsion_s"] = round(time.perf_counter() - ts, 2)
        ts = time.perf_counter()
        O.project_points(sph, moved, numf, proj_max_iters=3)
        stages["full_project_T3_sphere_s"] = round(time.perf_counter() - ts, 2)
        stages["full_points"] = PF
        # (3) SIREN projection, 100 k points
        ts = time.perf_counter()
        O.project_points(m, full[:, :100000], torch.tensor([100000]), proj_max_iters=10)
        stages["siren_project_T10_100k_s"] = round(time.perf_counter() - ts, 2)
    except Exception as e:                    # the per-stage block is informative; the baseline of record is (1)
        stages["full_size_error"] = repr(e)
    return {"value": round(P / t_all / 1e6, 6), "unit": "Mpoints/s", "cores": ncores, "kind": "port",
            "sample": "oracle (torch-CPU restatement of the reference's PyTorch path, %d threads; C rasteriser "
                      "single-threaded) on %d points, SIREN 4x256 fitted, project T=10 + resample + splat "
                      "fwd/bwd at %dx%dx%d view: %.1f s total, %.1f s project+resample"
                      % (ncores, P, S, S, V, t_all, t_pr),
            "stages": stages}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly the way the driver would
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    ...`), pass their output through and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:                          # a free port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def dry_run(args, world, rank):
    """ISO_BENCH_DRYRUN=1 (CPU test of the launch path, tests/test_dist_cpu.py): the ranks only rendezvous over gloo,
    run the timing protocol's barrier / max-over-ranks on an empty step and rank 0 prints a line marked as such."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks": t.item()}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))       # started bare: become the launcher of our own ranks
    if os.environ.get("ISO_BENCH_DRYRUN"):
        return dry_run(args, world, rank)
    if os.environ.get("ISO_BENCH_ONE_DEVICE"):     # test hook: all ranks on cuda:0 (with ISO_BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = os.environ.get("ISO_BENCH_BACKEND", "nccl")       # "nccl" = RCCL; gloo only for the 1-GPU dry run
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from iso_points_amd.dist import Comm
    comm = Comm(enabled=(world > 1))
    model = fitted_siren(dev)
    if dist is not None:
        # replicated weights = rank 0's (broadcast once per optimiser step); the per-rank
        # fits agree only up to the GEMM library's reduction order
        for prm in model.parameters():
            dist.broadcast(prm.data, src=0)
    cyc = Cycle(dev, model, comm)

    for _ in range(args.warmup):
        cyc.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    cyc.timed = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cyc.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cyc.timed = False
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3

    usage = checked(cyc, comm, dev)                      # untimed; raises if any capacity overflowed on any rank
    siren_ms, siren_launches = cyc.siren_stats()
    counts = cyc.active_counts()
    # ~2 s of back-to-back cycles outside the timed region, so that an external utilisation sampler (the driver's
    # rocm-smi samples) sees the GPU at work: the timed region itself lasts a fraction of a second
    # (a FIXED number of cycles: with N ranks every step holds collectives, so all ranks must run the same count -- a
    # wall-clock bound would let them disagree and hang)
    busy_steps = int(float(os.environ.get("ISO_BENCH_BUSY_S", "2.0")) / max(ms_per_step * 1e-3, 1e-4))
    busy_steps = comm.max_int(min(busy_steps, 2000), dev)
    for _ in range(busy_steps):
        cyc.step()
    torch.cuda.synchronize()
    evals_per_step = sum(sum(c) for c in counts)          # this rank's share
    flop_per_step = evals_per_step * FLOP_PER_EVAL
    launches_per_step = siren_launches / max(args.steps, 1)
    ach = flop_per_step / (siren_ms / args.steps * 1e-3) / 1e12 if siren_ms > 0 else 0.0

    from iso_points_amd import _lib
    x3 = _lib.load().iso_siren_get_gemm_mode() == 1
    # roofline peak for ALGORITHMIC (f32-equivalent) flops: the f32 MFMA peak for the f32 kernel;
    # for the split-operand kernel an algorithmic flop costs 3 fp16 MFMA flops, so the ceiling is the
    # fp16 dense peak / 3 = 839 TFLOP/s (executed fp16 flops = 3 x achieved, reported alongside).
    peak = PEAK_BF16_MFMA_TFLOPS / X3_PASSES if x3 else PEAK_F32_MFMA_TFLOPS
    traffic_bytes, traffic_src = measured_traffic(x3)
    if rank == 0:
        out = {
            "metric": "Mpoints/s full iso-point cycle (project+resample+splat), 1M pts",
            "value": round(P_TOTAL / (ms_per_step * 1e-3) / 1e6, 4),
            "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",            # 1 M points in total for every N (BASELINE.json: the same cycle at 1 and 8 GPUs)
            # no run of this command on more than one GPU exists in the builder's records (one GPU per lease): the N > 1 path
            # is verified bit-identical in lock-step / over gloo only; its compute ceiling: profiles/r05_rank_share_*.json
            # (true only when this very run has N > 1 ranks on N distinct GPUs over RCCL: tests/test_rccl_gpu.py)
            "scaling_measured": bool(world > 1 and not os.environ.get("ISO_BENCH_ONE_DEVICE")
                                     and os.environ.get("ISO_BENCH_BACKEND", "nccl") == "nccl"),
            "vs_baseline": None,
            "dtype": "f32-class (split-fp16 operands, 2^-22: hidden-layer products from two fp16 parts per f32 operand "
                     "under exact power-of-two scales, 3 fp16-MFMA passes, f32 accumulate; everything else f32)" if x3 else "f32",
            "data": "synthetic",
            "overflow": None,                # checked(): pair / row / halo capacities of the timed cycle held on every rank
            "config": {"workload": "configs[2]: 1M points project(T=10)+resample(FRNN K=9, repulsion, T=3) + EWA "
                                   "splat fwd/bwd 512x512x4 views, K=8",
                       "sdf": "SIREN 3->256x4->1 (omega 30), fitted to the unit sphere (300 Adam steps, seed 0)",
                       "points": P_TOTAL,
                       "point_order": "x-slab major, z-order curve inside a slab: one sort of the input cloud at set-up, "
                                      "untimed (dist.slab_order); `generator_order` below is the same cycle without it",
                       "parallelism": "1 rank" if world == 1 else
                       "x-slabs of the cloud (per-point stages; halo cells all-gathered) and tile-row bands "
                       "(per-pixel stages; each packed row sent to the bands it touches by ONE band all-to-all, its z sums "
                       "and visible flags returned by the reverse all-to-all) x%d ranks, RCCL" % world},
            "roofline": {"bound": "mfma",
                         "kernel": ("k_siren_step_x3_both<256,8,3,1> (fused SIREN SDF+grad Newton step, split-fp16 MFMA; 96- and 32-point tiles of a list in one launch) + k_siren_tail_x3 (the late iterations of a projection in one launch)" if x3
                                    else "k_siren_step<16> (fused SIREN SDF+grad Newton step, f32 MFMA)"),
                         "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4),
                         "frac_of_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         # HBM-side bytes per launch, NOT measured in this run (bench.py cannot profile itself): PMC
                         # passes of this same command in the builder's last profiled run (see measured_traffic)
                         "traffic": traffic_bytes,
                         "traffic_source": traffic_src,
                         "traffic_note": "bytes/launch of the dominant kernel from the PMC passes (2*FETCH_SIZE+WRITE_SIZE) of "
                                         "an earlier run of this command, file named in traffic_source -- not this run; "
                                         "algorithmic point I/O is %.1f MB/launch, the rest is the w*cos stash round trip"
                                         % (evals_per_step / max(launches_per_step, 1) * 37 / 1e6),
                         "peak_note": ("fp16 dense MFMA peak 2516.6 / 3 passes per f32 product; executed fp16 "
                                       "rate = %.1f TFLOP/s" % (ach * X3_PASSES)) if x3
                         else "f32 dense MFMA peak",
                         # what the part sustains: with all 256 CUs issuing dense fp16 MFMAs on non-trivial operands the
                         # shader clock settles at 1.66 GHz, not the 2.4 GHz the peak assumes (tools/probes/clock.hip,
                         # profiles/r02_clock_probe.txt); frac_of_sustained prices the executed rate against that
                         "sustained_peak": round(SUSTAINED_FP16_TFLOPS / X3_PASSES, 1) if x3 else None,
                         "frac_of_sustained": round(ach / (SUSTAINED_FP16_TFLOPS / X3_PASSES), 4) if x3 else None,
                         # ... and on operands like the kernel's own (noise-like fp16 halves, new ones for every MFMA)
                         "frac_of_sustained_random_operands": round(ach / (SUSTAINED_FP16_RANDOM_TFLOPS / X3_PASSES), 4) if x3 else None,
                         "launches_per_step": launches_per_step,
                         "avg_launch_ms": round(siren_ms / max(siren_launches, 1), 4),
                         "point_evals_per_step_rank0": evals_per_step,
                         "active_points_per_launch_rank0": counts,
                         "share_of_step": round(siren_ms / args.steps / ms_per_step, 4)},
        }
        if world == 1:
            out["cfg3a_analytic_sdf"] = analytic_cycle(dev, comm, args)
            out["f32_mfma_mode"] = f32_mode_cycle(dev, model, comm)
            out["generator_order"] = generator_order_cycle(dev, model, comm)
            out["operator_api"] = operator_api_cycle(dev, model)
            out["operator_api"]["vs_headline"] = round(out["operator_api"]["ms_per_step"] / ms_per_step, 3)
            out["operator_api_small"] = operator_api_small(dev, model)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
