"""How many K=7 / K=9 queries of the bench cycle end up in the tail kernel, and why."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from iso_points_amd import frnn
from iso_points_amd.dist import Comm
from iso_points_amd.sdf_models import SphereSDF
dev = torch.device("cuda:0")
cyc = bench.Cycle(dev, SphereSDF().to(dev), Comm(enabled=False)).cyc
r = cyc.project_resample()
orig = frnn.frnn_grid_points
def spy(*a, **k):
    out = orig(*a, **k)
    g = out[3]
    torch.cuda.synchronize()
    print("query K=%s P1=%s: tail counts %s, grid res %s cell %.5f" % (k.get("K"), a[0].shape[1], g.tail_counts.tolist(),
          g.params[:, 4:7].tolist()[0], 1.0 / float(g.params[0, 3])))
    return out
frnn.frnn_grid_points = spy
import iso_points_amd.rasterizer as R
R.frnn.frnn_grid_points = spy
frags, filt = cyc.splat_forward(r.points[0], r.normals[0])
