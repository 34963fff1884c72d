import sys; sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))),'tests')); sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import test_idr_gpu as T
from iso_points_amd.levelset_sampling import UniformProjection, full_lengths
from iso_points_amd.sdf_models import idr_sdf_and_grad
dev=torch.device('cuda:0')
g=T.load("idr_small.npz"); m=T.idr_from(g)
sdf,grad=idr_sdf_and_grad(m,g["points"].to(dev))
print("sdf",T.rel_err(sdf,g["sdf"]),"grad",T.rel_err(grad,g["grad"]))
x=g["points"].to(dev)
r=UniformProjection(proj_tolerance=1e-30)._project_points(m,x,full_lengths(x),proj_max_iters=int(g["T"]))
print("pts",T.rel_err(r.points,g["fixed_points"]))
ne=(r.normals.cpu()-g["fixed_normals"]).abs().amax(-1)/g["fixed_normals"].abs().max()
print("ne median %.3g frac>1e-4 %.4f max %.3g n=%d"%(ne.median(),(ne>1e-4).float().mean(),ne.max(),ne.numel()))
