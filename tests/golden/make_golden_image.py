"""Golden vectors for the image look-up, made by the REFERENCE's own
get_tensor_values (DSS/utils/__init__.py:325-375) imported through make_golden.py's shims.
usage:  ISO_GOLDEN_ONLY=image python tests/golden/make_golden.py"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
for _p in (_HERE, _os.path.dirname(_os.path.dirname(_HERE))):      # make_golden.py and the repo root (oracle/)
    if _p not in _sys.path:
        _sys.path.insert(0, _p)

import sys

import torch

from make_golden import npz


def gen_image(L):
    U = sys.modules["DSS.utils"]
    g = torch.Generator().manual_seed(21)
    # a soft-edged mask (1 channel) and a colour image (3 channels), non-square
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 48), torch.linspace(-1, 1, 64), indexing="ij")
    mask = ((xx * xx + yy * yy) < 0.5).float().view(1, 1, 48, 64).repeat(2, 1, 1, 1)
    mask[1] = ((xx.abs() + yy.abs()) < 0.8).float()
    rgb = torch.rand(2, 3, 48, 64, generator=g)
    # samples inside, on the border (+-1 exactly) and outside (reflection padding) of [-1, 1]
    p = (torch.rand(2, 3000, 2, generator=g) - 0.5) * 2.6
    p[:, :8] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [-1.0, 1.0], [1.0, -1.0], [0.0, 0.0], [1.0, 0.0],
                             [-1.3, 0.2], [2.9, -3.1]])
    out = {"mask": mask, "rgb": rgb, "p": p}
    out["mask_bilinear"], out["mask_valid"] = U.get_tensor_values(mask, p.clone(), with_mask=True, squeeze_channel_dim=True)
    out["rgb_bilinear"] = U.get_tensor_values(rgb, p.clone())
    out["rgb_nearest"] = U.get_tensor_values(rgb, p.clone(), mode="nearest")
    sq = torch.rand(2, 3, 40, 40, generator=g)
    pin = (torch.rand(2, 500, 2, generator=g) - 0.5) * 2.0
    out["sq"], out["p_in"] = sq, pin
    out["sq_index"] = U.get_tensor_values(sq, pin.clone(), grid_sample=False)
    npz("image_values.npz", **out)

if __name__ == "__main__":          # this part alone: python tests/golden/make_golden_image.py
    _os.environ["ISO_GOLDEN_ONLY"] = "image"
    import make_golden
    make_golden.main()
