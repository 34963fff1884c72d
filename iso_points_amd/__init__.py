"""iso_points_amd -- MI355X (gfx950) implementation of the iso-point hot path of
yifita/iso-points: Newton level-set projection, FRNN grid search, tangent-plane
repulsion resample and EWA surface splatting (forward + backward), behind the
reference's own operator signatures.  All compute lives in libisopoints_hip.so
(C ABI: include/isopoints.h); this package is the thin host-side mirror.

(The directory `iso-points_amd` at the repo root is a symlink to this package:
a hyphen is not importable in Python.)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "frnn", "prefix_sum", "levelset_sampling", "sdf_models"]
