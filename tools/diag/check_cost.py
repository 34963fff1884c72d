"""What watching the capacities costs: the headline cycle with IsoCycle.check() (host reads of the device-side usage
counters: one synchronisation + a few small copies) after EVERY step, against the cycle alone; and one calibrate().
usage: python tools/diag/check_cost.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd.dist import Comm
dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, bench.fitted_siren(dev), Comm(enabled=False))
for _ in range(3):
    out = cyc.step()
torch.cuda.synchronize()


def run(n, with_check):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = cyc.step()
        if with_check:
            cyc.cyc.check(out[4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = run(20, False); b = run(20, True); a2 = run(20, False)
print("cycle alone %.3f / %.3f ms, with check() after every step %.3f ms (+%.3f)" % (a, a2, b, b - 0.5 * (a + a2)))
t0 = time.perf_counter(); cyc.cyc.calibrate(); torch.cuda.synchronize()
print("one calibrate(): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
