"""The operator-API cycle of bench.py alone (timing + a torch-profiler style breakdown by wall clock).
usage: python tools/opapi_only.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
print(bench.operator_api_cycle(dev, model, steps=3))
