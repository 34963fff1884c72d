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
