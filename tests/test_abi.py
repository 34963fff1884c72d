"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every
symbol include/isopoints.h declares, and the ctypes table covers exactly that set.
No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "isopoints.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(iso_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "iso_project_sphere" in syms and "iso_frnn_query" in syms and len(syms) > 10


def test_library_exports_every_declared_symbol():
    from iso_points_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first (make / __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in isopoints.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from iso_points_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()
    lib = _lib.load()
    assert lib.iso_version().decode().startswith("isopoints-hip")


def test_size_helpers_need_no_gpu():
    from iso_points_amd import _lib
    lib = _lib.load()
    H, L = 256, 3
    assert lib.iso_siren_raw_floats(H, L) == H * 3 + H + L * (H * H + H) + H + 1
    # f32-MFMA images + K-order vectors + split-fp16 images (header, forward, transposed)
    assert lib.iso_siren_packed_floats(H, L) == (5 * H + 4 + L * (H + 2 * H * H)) + (5 * H + L * H) + (24 + 2 * L * H * H)
    assert lib.iso_prefix_sum_workspace_bytes(1, 1) >= 4
    assert lib.iso_project_siren_workspace_bytes(1000, H, L) > 0


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from iso_points_amd import frnn
    x = torch.rand(1, 10, 3)
    with pytest.raises(RuntimeError):
        frnn.frnn_grid_points(x, x, K=3, r=0.5)


def test_no_vgpr_spills_in_the_raster_kernels():
    """Code-object metadata of the built library (tools/spill_check.py): the candidate-parallel raster kernels of the cycle
    (K <= 8) must not spill vector registers -- a build of k_raster<8, true> with two spilled VGPRs returned stale list
    entries for pixels with depth ties (profiles/HISTORY.md, round 5)."""
    import importlib.util
    import shutil
    import pytest
    if not shutil.which("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no ROCm llvm tools")
    from iso_points_amd import _lib
    spec = importlib.util.spec_from_file_location("spill_check", os.path.join(ROOT, "tools", "spill_check.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    ks = sc.kernels(_lib.LIB_PATH)
    assert len(ks) > 100, "code objects not found in the library"
    raster = [k for k in ks if "8k_rasterILi4ELb1ELb" in k[0] or "8k_rasterILi8ELb1ELb" in k[0]]
    assert len(raster) == 4, [k[0] for k in ks if "raster" in k[0]]
    assert all(k[1] == 0 for k in raster), raster
