"""bench.operator_api_small alone (the reference's working set: 24 000 points, one view).  usage: python tools/opapi_small.py [P]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
model = bench.fitted_siren(dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
print(json.dumps(bench.operator_api_small(dev, model, P=P)))
