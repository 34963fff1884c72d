"""The operator-API cycle of bench.py alone.  usage: python tools/opapi_only.py [steps] [trace]
With `trace` every step is bracketed by a marker kernel (torch.cuda._sleep -> spin_kernel) so that
tools/opapi_sequence.py can cut one step out of a rocprofv3 kernel trace."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if "trace" in sys.argv[2:]:
    bench.OPAPI_STEP_HOOK = lambda: torch.cuda._sleep(2000)
print("operator API cycle:", bench.operator_api_cycle(dev, model, steps=steps))
