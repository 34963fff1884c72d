"""The arithmetic behind the split-fp16 SIREN kernel, emulated with numpy on the CPU:
an f32 operand cut into two fp16 numbers under a power-of-two scale is represented to the f32
rounding level, and three partial products reproduce W.x as well as an f32 GEMM does.  (The kernel
itself is checked on the GPU; this pins the error analysis the design rests on.)"""
import numpy as np
import torch


def split_f16(x):
    h = x.astype(np.float16).astype(np.float32)
    lo = (x - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, lo


def pow2_scale(mx, target_exp):
    """2^k with mx * 2^k in [2^(target_exp-1), 2^target_exp) (siren_x3.hip: k_siren_wscale / x3_scale_for)."""
    _, e = np.frexp(np.float32(mx))
    return np.float32(2.0) ** np.float32(target_exp - e)


def test_two_part_cut_is_exact_to_f32_rounding():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-3, 3, 200000))).astype(np.float32)
    s = pow2_scale(np.abs(x).max(), 12)
    h, lo = split_f16(x * s)
    rel = np.abs((h.astype(np.float64) + lo) / s - x) / np.maximum(np.abs(x), 1e-30)
    big = np.abs(x * s) >= 2.0 ** -2            # low part is a normal fp16 number
    assert rel[big].max() <= 2.0 ** -23
    # below that the cut has fixed absolute resolution 2^-25 in scaled units (fp16 subnormals)
    assert np.abs((h.astype(np.float64) + lo) - x * s).max() <= 2.0 ** -13 * 2.0 ** -11 * 4096


def test_three_products_match_an_f32_gemm():
    rng = np.random.default_rng(1)
    H = 256
    for wmax, xgen in ((np.sqrt(6 / H) / 30, lambda n: np.sin(rng.normal(size=(n, H)) * 3)),
                       (0.3, lambda n: np.sin(rng.normal(size=(n, H)) * 3)),
                       (np.sqrt(6 / H) / 30, lambda n: np.sin(rng.normal(size=(n, H)) * 0.05))):
        W = rng.uniform(-wmax, wmax, size=(H, H)).astype(np.float32)
        X = xgen(1500).astype(np.float32)
        ref = X.astype(np.float64) @ W.astype(np.float64).T
        f32 = (torch.from_numpy(X) @ torch.from_numpy(W).T).numpy()
        sw = pow2_scale(np.abs(W).max(), 10)
        wh, wl = split_f16(W * sw)
        xh, xl = split_f16(X * np.float32(4096.0))
        acc = np.zeros_like(ref, dtype=np.float32)
        for a, b in ((wl, xh), (wh, xl), (wh, xh)):          # the kernel's order: smallest terms first
            acc = (acc + b.astype(np.float64) @ a.astype(np.float64).T).astype(np.float32)
        got = acc.astype(np.float64) / (sw * 4096.0)
        scale = np.abs(ref).max()
        e_split = np.sqrt(((got - ref) ** 2).mean()) / scale
        e_f32 = np.sqrt(((f32 - ref) ** 2).mean()) / scale
        # operand truncation of the split is below (typically 4x below) the accumulation error of an f32 GEMM
        assert e_split <= 1.5 * e_f32, (e_split, e_f32)
        assert np.abs(got - ref).max() / scale <= 1e-6


def test_adjoint_bound_never_overflows_fp16():
    """|(W^T a)[f] * w cos| <= w * max_f sum_k |W[k][f]| * max|a|: scaled to below 2^14 by that bound the
    next adjoint always fits fp16, whatever the weights."""
    rng = np.random.default_rng(2)
    H = 128
    for wmag in (1e-3, 0.05, 30.0):
        W = (rng.standard_normal((H, H)) * wmag).astype(np.float32)
        a = (rng.standard_normal((500, H)) * np.exp(rng.uniform(-8, 8, (500, 1)))).astype(np.float32)
        w = np.float32(30.0)
        nxt = (a.astype(np.float64) @ W.astype(np.float64)) * w * rng.uniform(-1, 1, (500, H))
        bound = w * np.abs(W).sum(axis=0).max() * np.abs(a).max(axis=1) * 1.01
        scale = np.array([pow2_scale(b, 14) for b in bound], dtype=np.float64)
        assert (np.abs(nxt) * scale[:, None]).max() < 2.0 ** 14 < 65504
