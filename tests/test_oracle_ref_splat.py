"""Pin oracle_splat.c against the REFERENCE's own CPU rasteriser
(DSS/csrc/rasterize_points_cpu.cpp compiled as-is into oracle/_ref/ by oracle/Makefile).
Bit-exact: integer index lists and float32 values."""
import pytest
import torch

from oracle import splat_oracle as SO
from splat_util import random_splats, sphere_scene

needs_ref = pytest.mark.skipif(not SO.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _fwd_both(sc, S, K, thres=0.05):
    args = (sc["ndc"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first"], sc["num"], thres, S, K)
    return SO.splat_forward(*args, use_ref=True), SO.splat_forward(*args, bbox_or=False)


@needs_ref
@pytest.mark.parametrize("K", [1, 5, 8])
def test_forward_matches_reference_cpu_random(K):
    sc = random_splats(400, N=2, seed=K)
    ref, ours = _fwd_both(sc, 32, K)
    for r, o in zip(ref, ours):
        assert torch.equal(r, o)
    assert (ref[0] >= 0).any()


@needs_ref
def test_forward_matches_reference_cpu_scene():
    sc = sphere_scene(3000, n_views=2, S=48, seed=1)
    ref, ours = _fwd_both(sc, 48, 8)
    for r, o in zip(ref, ours):
        assert torch.equal(r, o)
    assert ref[3].sum() > 100


@needs_ref
def test_cuda_reject_rule_equals_cpu_rule_when_radii_bound_the_ellipse():
    """`||` (CUDA, canonical) and `&&` (CPU) agree whenever radii contain the true bbox of Q<=C
    (SURVEY 8(a) divergence note) -- padded by 1e-3 here to stay clear of rounding."""
    sc = random_splats(400, N=1, seed=9, pad=1.001)
    args = (sc["ndc"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first"], sc["num"], 0.05, 40, 6)
    ref = SO.splat_forward(*args, use_ref=True)
    cuda_rule = SO.splat_forward(*args, bbox_or=True)
    for r, o in zip(ref, cuda_rule):
        assert torch.equal(r, o)


@needs_ref
def test_occ_backward_cpu_mode_matches_reference():
    sc = sphere_scene(1500, n_views=2, S=32, seed=2)
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(2, 32, 32, generator=g)
    grad[grad.abs() < 0.6] = 0.0
    ref = SO.occ_backward(sc["ndc"], sc["radii"], grad, sc["first"], sc["num"], 10.0, mode=0, use_ref=True)
    ours = SO.occ_backward(sc["ndc"], sc["radii"], grad, sc["first"], sc["num"], 10.0, mode=0)
    assert torch.equal(ref, ours)
    assert ref.abs().sum() > 0


@needs_ref
def test_zbuf_backward_matches_reference():
    sc = sphere_scene(1500, n_views=2, S=32, seed=4)
    idx, zb, qv, occ = SO.splat_forward(sc["ndc"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first"],
                                        sc["num"], 0.05, 32, 5)
    g = torch.Generator().manual_seed(5)
    gz = torch.randn(zb.shape, generator=g)
    gz[gz.abs() < 0.3] = 0
    P = sc["ndc"].shape[0]
    assert torch.equal(SO.zbuf_backward(idx, gz, P, use_ref=True), SO.zbuf_backward(idx, gz, P))


def test_backward_modes_are_consistent_without_reference():
    """mode 1 (CUDA rect) vs mode 0 (CPU rect): same support whenever a pixel is inside the scaled
    rect in both axes or outside in both; mode 2 (disc) is a different support by design."""
    sc = sphere_scene(800, n_views=1, S=24, seed=6)
    g = torch.Generator().manual_seed(7)
    grad = -torch.rand(1, 24, 24, generator=g)      # all negative: no 'outside splat' skipping
    m1 = SO.occ_backward(sc["ndc"], sc["radii"], grad, sc["first"], sc["num"], 1000.0, mode=1)
    m0 = SO.occ_backward(sc["ndc"], sc["radii"], grad, sc["first"], sc["num"], 1000.0, mode=0)
    m2 = SO.occ_backward(sc["ndc"], sc["radii"], grad, sc["first"], sc["num"], 1000.0,
                         rs=torch.tensor([100.0]), mode=2)
    assert torch.allclose(m0, m1, rtol=1e-6, atol=1e-6) and torch.allclose(m1, m2, rtol=1e-6, atol=1e-6)
