"""First contact with real multi-GPU hardware: everything here needs >= 2 visible GPUs and is
SKIPPED otherwise (every lease so far had one).  One command on such a node:

    python -m pytest tests/test_rccl_gpu.py -m gpu -q

(1) the collectives dist.Comm issues (all_gather_into_tensor, all_to_all_single with equal splits, all_reduce sum / max)
    over RCCL ("nccl") against their definitions;
(2) the sharded cycle under RCCL, one rank per GPU: bit-identical to the single-GPU cycle (the same worker the gloo
    test runs on one GPU);
(3) bench.py --gpus N through torch.distributed.run as the driver launches it: the contract's line with
    "scaling_measured": true."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


needs2 = pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs (RCCL over xGMI); %d visible" % _n_gpus())

_COLL = r"""
import os, sys, torch
import torch.distributed as dist
dist.init_process_group(backend="nccl")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
sys.path.insert(0, sys.argv[1])
from iso_points_amd.dist import Comm
c = Comm()
x = torch.arange(5, device=dev, dtype=torch.float32) + 10 * rank
g = c.execute(("all_gather", x))
ok = all(torch.equal(g[r], torch.arange(5, device=dev, dtype=torch.float32) + 10 * r) for r in range(world))
a = (torch.arange(world * 3, device=dev, dtype=torch.float32).view(world, 3) + 100 * rank)
t = c.execute(("all_to_all", a))
ok = ok and all(torch.equal(t[s], torch.arange(rank * 3, rank * 3 + 3, device=dev, dtype=torch.float32) + 100 * s) for s in range(world))
s_ = c.execute(("all_reduce", torch.full((4,), float(rank + 1), device=dev), "sum"))
ok = ok and torch.equal(s_, torch.full((4,), world * (world + 1) / 2.0, device=dev))
m = c.execute(("all_reduce", torch.tensor([rank, -rank], dtype=torch.int32, device=dev), "max"))
ok = ok and m.tolist() == [world - 1, 0]
torch.cuda.synchronize()
print("RANK", rank, "OK" if ok else "MISMATCH")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def _run(script_text, tmp_path, n, port, extra=()):
    w = tmp_path / "worker.py"
    w.write_text(script_text)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(w), ROOT] + list(extra),
                          capture_output=True, text=True, timeout=900, env=env)


@needs2
def test_comm_collectives_over_rccl(tmp_path):
    n = min(_n_gpus(), 8)
    out = _run(_COLL, tmp_path, n, 29541)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("OK") == n


@needs2
def test_sharded_cycle_over_rccl_equals_single_gpu(tmp_path):
    from test_dist_gpu import _WORKER
    n = min(_n_gpus(), 8)
    # the gloo worker with RCCL and one GPU per rank
    text = _WORKER.replace('dist.init_process_group(backend="gloo")',
                           'dist.init_process_group(backend="nccl"); torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))') \
                  .replace('dev = torch.device("cuda:0")', 'dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))')
    assert "nccl" in text and "LOCAL_RANK" in text
    out = _run(text, tmp_path, n, 29543)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("OK") == n


@needs2
def test_bench_over_rccl_prints_a_measured_scaling_line():
    n = 2 if _n_gpus() < 4 else (4 if _n_gpus() < 8 else 8)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["value"] > 0
    assert d.get("scaling_measured") is True, "a run on %d real GPUs must say so" % n
    print("bench over RCCL, %d GPUs: %.2f Mpoints/s, %.3f ms per step" % (n, d["value"], d["ms_per_step"]))
