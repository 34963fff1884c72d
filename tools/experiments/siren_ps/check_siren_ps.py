"""Experiment check (not part of tests/): tools/experiments/siren_ps/build.sh first, then
    python -m pytest tools/experiments/siren_ps/check_siren_ps.py -m gpu -q
The point-stationary form of the SIREN step (siren_ps.hip in the VARIANT library, ISO_SIREN_PS=1) against k_siren_step_x3:
same split-fp16 arithmetic in the same order, so evaluations (value + gradient, ragged list lengths, L = 2 and 3) and a
T = 10 projection (device-side lists, moves, compaction) must agree BIT FOR BIT.  The kernel is selected once per process
from the environment, hence two subprocesses of ps_check.py."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANT = os.path.join(ROOT, "tools", "variants", "libiso_siren_ps.so")


@pytest.mark.gpu
def test_point_stationary_step_is_bit_identical_to_the_feature_split_step(tmp_path):
    outs = []
    for flag in ("0", "1"):
        out = str(tmp_path / ("ps_%s.pt" % flag))
        if not os.path.exists(VARIANT):
            pytest.skip("build the variant first: tools/experiments/siren_ps/build.sh")
        env = dict(os.environ, ISO_SIREN_PS=flag, PS_CHECK_SMALL="1", ISO_DEV_LIB=VARIANT)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "experiments", "siren_ps", "ps_check.py"), "run", out], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out)
    a, b = torch.load(outs[0]), torch.load(outs[1])
    assert sorted(a) == sorted(b) and len(a) > 20
    for k in sorted(a):
        x, y = a[k], b[k]
        if x.is_floating_point():
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), k
        else:
            assert torch.equal(x, y), k
