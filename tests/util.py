"""Shared helpers for the parity tests."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def sphere_cloud(P, seed=0, jitter=0.05):
    """SURVEY 8(d) cfg 2/3 input: normalize(randn) + jitter*(rand-0.5)."""
    g = torch.Generator().manual_seed(seed)
    p = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1)
    return p + jitter * (torch.rand(1, P, 3, generator=g) - 0.5)


def cube_cloud(P, seed=0):
    """SURVEY 8(d) cfg 1 input: (rand-0.5)*2."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(1, P, 3, generator=g) - 0.5) * 2


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


FLIP_LOG = []


def assert_projection_close(res_pts, ref_pts, stop_tol=5e-5, tol=1e-5, max_flip_frac=1e-3):
    """Positions after Newton projection.  Every point must agree to `tol` relative,
    except "stop flips": a point whose |sdf| lands within rounding of the stopping
    tolerance can take one move more or less on one side; such a point may differ by at
    most ~stop_tol (one residual Newton move), and they must be rare: at most 0.1 % of the cloud
    (two points for clouds below 2000: one flip in a 659-point cloud is 0.15 %).  The count is printed
    (pytest -s) and kept in FLIP_LOG; measured over the GPU suite: 0 in most calls, at most 2 of 2000.
    Tests that pin the iteration count (stop_tol ~ 0) check the strict bound on every point."""
    a, b = res_pts.detach().cpu().double(), ref_pts.detach().cpu().double()
    scale = b.abs().max().clamp_min(1e-30)
    err = (a - b).abs().amax(dim=-1) / scale
    bad = err > tol
    frac = bad.double().mean().item() if bad.numel() else 0.0
    FLIP_LOG.append((frac, int(bad.sum().item()), int(bad.numel())))
    print("assert_projection_close: %d of %d points (%.4f%%) beyond %g (stop flips)" % (bad.sum().item(), bad.numel(), 100 * frac, tol))
    assert bad.sum().item() <= max(max_flip_frac * bad.numel(), 2), \
        "%.4f%% of points differ by more than %g (stop flips must stay below %.2f%%)" % (100 * frac, tol, 100 * max_flip_frac)
    if bad.any():
        assert err[bad].max().item() < 3 * stop_tol / scale.item() + tol, \
            "point differs by %g (> one residual Newton move)" % err[bad].max().item()


_FIT_CACHE = {}


def fitted_siren(O, hidden, n_layers, seed=0, fit=0):
    """oracle SirenSDF(hidden, n_layers) seeded and Adam-fitted to the unit sphere for `fit` steps.
    The (deterministic, CPU) fit takes ~20 s for 4x256: it is done once per test session and a
    deep copy is handed out."""
    import copy
    key = (hidden, n_layers, seed, fit)
    if key not in _FIT_CACHE:
        torch.manual_seed(seed)
        m = O.SirenSDF(hidden_size=hidden, n_layers=n_layers)
        if fit:
            O.fit_siren_to_sphere(m, steps=fit)
        _FIT_CACHE[key] = m
    return copy.deepcopy(_FIT_CACHE[key])
