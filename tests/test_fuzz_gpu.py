"""A short run of the two fuzzers under tools/ (hundreds of cases each were run on an MI355X while round 4 was built;
`python tools/fuzz_fused.py 400` / `python tools/fuzz_splat.py 120` repeat that): adversarial clouds through the fused
neighbour kernels against the stand-alone FRNN / repulsion / bandwidth path, random splat sets through the forward and
backward splat kernels against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fused_neighbour_kernels_on_adversarial_clouds(dev):
    assert _load("fuzz_fused").run(21, 1000) == 0


def test_splat_kernels_on_random_splat_sets(dev):
    assert _load("fuzz_splat").run(28, 2000) == 0
