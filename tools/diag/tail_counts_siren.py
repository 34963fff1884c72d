"""Tail-query counts per cycle of the SIREN (headline) cycle.  usage: python tools/diag/tail_counts_siren.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from iso_points_amd.dist import Comm
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
cyc = bench.Cycle(dev, model, Comm(enabled=False))
cyc.cyc.use_graphs = False
cyc.step()
cyc.cyc.grid.counters_since_last()
out = cyc.step()
u = cyc.cyc.usage(out[4])
print("per cycle:", {k: u["grid"][k] for k in ("occupied", "tail", "tail_h", "overflow_bricks")})
