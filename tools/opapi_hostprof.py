"""Host-side profile (cProfile) of the operator-API step: where the Python time goes while the GPU waits.
usage: python tools/opapi_hostprof.py"""
import cProfile, os, pstats, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
pr = cProfile.Profile()
state = {"n": 0}
def hook():
    state["n"] += 1
    if state["n"] == 3:
        pr.enable()
bench.OPAPI_STEP_HOOK = hook
r = bench.operator_api_cycle(dev, model, steps=8)
pr.disable()
print(r["ms_per_step"], r["ms_per_step_min_max"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
