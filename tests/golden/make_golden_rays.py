"""Golden vectors for the ray-side sampling of CombinedModel: the reference's own
statements DSS/models/combined_modeling.py:324-386 (ray -> nearest frontal / occluded iso-point over the
dense (R,M) matrices, the in-surface segment, the lowest-SDF candidate on it) are read from the checkout AT
GENERATION TIME and executed on synthetic inputs; only the inputs and results are stored.  The method
cannot be called as a whole here (it needs pytorch3d cameras and the renderer); everything these lines
touch is supplied by plain stand-ins below and by the reference's own helpers (DSS.utils: eps_sqrt,
gather_batch_to_packed, num_points_2_packed_to_cloud_idx)."""
import os
import sys
import textwrap
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIRST, LAST = 324, 386          # "# TODO: faster search" ... "p_insurface = torch.gather(...).squeeze(-2)"


class PL(object):
    def __init__(self, lst):
        self._l = lst

    def points_list(self):
        return self._l


def gen_rays():
    from make_golden import install_shims, npz
    install_shims()
    sys.path.insert(0, REF)
    import importlib
    import torch.nn.functional as F
    U = importlib.import_module("DSS.utils")
    MH = importlib.import_module("DSS.utils.mathHelper")
    from oracle import iso_oracle as O
    src = open(os.path.join(REF, "DSS", "models", "combined_modeling.py")).read().split("\n")[FIRST - 1:LAST]
    assert src[0].strip().startswith("# TODO: faster search") and "p_insurface = torch.gather" in "\n".join(src[-3:])
    text = textwrap.dedent("\n".join(src))
    # torch >= 2 rejects a masked write whose mask aliases the written tensor (the same incompatibility as
    # levelset_sampling.py:328, see make_golden.py): the mask is cloned, in memory, nothing else changes
    a = "mask_insurface[b][mask_insurface[b]] = "
    assert text.count(a) == 1
    text = text.replace(a, "mask_insurface[b][mask_insurface[b].clone()] = ")
    code = compile(text, "combined_modeling.py[%d:%d]" % (FIRST, LAST), "exec")

    def sphere_cloud(P, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    B, R, n_ray = 2, 700, 32
    cam_pos = torch.tensor([[0.0, 0.3, 3.0], [2.5, 0.0, 1.2]])
    frontal, occluded, samples = [], [], []
    for b in range(B):
        cloud = sphere_cloud(6000 + 500 * b, 70 + b)
        toward = (cloud * torch.nn.functional.normalize(cam_pos[b], dim=0)).sum(-1)
        frontal.append(cloud[toward > 0.1].contiguous())
        occluded.append(cloud[toward < -0.1].contiguous())
        g = torch.Generator().manual_seed(80 + b)
        samples.append(0.9 * (torch.rand(R, 3, generator=g) - 0.5) * 2)          # sample points inside the unit cube
    P_pad = R
    sample_points_padded = torch.stack(samples)
    mask_insurface = torch.ones((B, P_pad), dtype=torch.bool)
    model = O.SphereSDF(radius=0.8)
    ns = {
        "torch": torch, "F": F, "eps_sqrt": MH.eps_sqrt, "gather_batch_to_packed": U.gather_batch_to_packed,
        "num_points_2_packed_to_cloud_idx": U.num_points_2_packed_to_cloud_idx,
        "batch_size": B, "occluded_points": PL(occluded), "frontal_points": PL(frontal),
        "sample_points_lst": [s.clone() for s in samples], "cam_pos": cam_pos.clone(),
        "mask_insurface": mask_insurface, "sample_points_padded": sample_points_padded,
        "n_points_per_ray": n_ray, "lengths": torch.zeros(1),
        "self": types.SimpleNamespace(max_points_per_pass=100000, decoder=model),
    }
    exec(code, ns)
    out = {"cam_pos": cam_pos, "samples": sample_points_padded, "n_points_per_ray": n_ray, "sdf_radius": 0.8,
           "mask_insurface": ns["mask_insurface"], "ray_len0": ns["ray_len0"].reshape(-1),
           "ray_len1": ns["ray_len1"].reshape(-1), "num_ins_per_batch": ns["num_ins_per_batch"],
           "p_insurface": ns["p_insurface"], "cam_ray": ns["cam_ray"]}
    for b in range(B):
        out["frontal%d" % b] = frontal[b]
        out["occluded%d" % b] = occluded[b]
    npz("ray_sampling.npz", **out)


if __name__ == "__main__":
    gen_rays()
