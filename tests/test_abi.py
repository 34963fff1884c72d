"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every
symbol include/isopoints.h declares, and the ctypes table covers exactly that set.
No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re
import sys

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


def _code_object_kernels():
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
    return ks


def test_selection_list_files_are_built_without_the_slp_vectoriser():
    """Round 5 met a build of k_raster<8, true> (merge folded in, the slice's own K-best list carried in registers into the
    swap-chain insertion) that returned wrong lists on depth ties, a different set of pixels every run, and blamed two
    spilled registers.  Round 6 took it apart (tools/probes/spill_kit, profiles/r06_spill_repro_*.txt, profiles/HISTORY.md):
    the spilled registers are innocent (a build with none fails, a build with three is correct), the run-to-run variation
    is the tile bins' atomic order cutting a tile's candidates into different slices every run, and the defect is a
    deterministic miscompile by the SLP vectoriser of this toolchain: a swap taken by the TIE rule of the (z, id, q)
    insertion gets the values of the no-swap path when the list is carried around a loop.  tools/probes/tie_merge.hip
    reproduces it in 150 lines; -fno-slp-vectorize cures it there and in the failing raster builds.  So the guard is the
    build flag on every file that holds a selection list -- every file but siren_x3.hip -- plus tests/test_ties_gpu.py."""
    import subprocess
    out = subprocess.run(["make", "-n", "-B", "-C", ROOT, "iso_points_amd/libisopoints_hip.so"], stdout=subprocess.PIPE, text=True).stdout
    lines = [l for l in out.splitlines() if " -c " in l and ".hip" in l]
    assert len(lines) >= 16, out[-2000:]
    for l in lines:
        src = [w for w in l.split() if w.endswith(".hip")][0]
        base = os.path.basename(src)[:-4]
        if base in ("siren_x3",):
            txt = open(os.path.join(ROOT, src)).read()
            assert "med3" not in txt and ".push(" not in txt, base       # the files that keep SLP hold no selection list
        else:
            assert "-fno-slp-vectorize" in l.split(), l


# VGPR spills that are known and pinned by a repeat-stress test at full size (tests/test_round6_gpu.py); the number is the
# ceiling a rebuild may not exceed without a look.  Everything else in the library must not spill.
SPILL_ALLOWED = {
    "20k_siren_step_x3_bothILi256ELi8ELi3ELi1E": (99, "test_siren_step_repeat_stress_1m"),
    "15k_siren_step_x3ILi256ELi8ELi3ELi1ELb0E": (91, "test_siren_step_repeat_stress_1m"),
    "14k_idr_step_x16ILi512ELi2ELb0E": (100, "test_idr_step_repeat_stress_1m"),
    "14k_idr_step_x16ILi512ELi2ELb1E": (32, "test_idr_step_repeat_stress_1m"),
    "14k_idr_step_x16ILi256ELi3ELb0E": (83, "test_idr_step_repeat_stress_1m"),
    "14k_idr_step_x16ILi256ELi3ELb1E": (20, "test_idr_step_repeat_stress_1m"),
    "10k_idr_stepILi8E": (2, "test_idr_step_repeat_stress_1m"),
    "10k_idr_stepILi16E": (232, "test_idr_step_repeat_stress_1m"),
    "10k_fps_lazyILi8E": (2, "test_fps_repeat_stress_500k"),
    "10k_fps_lazyILi16E": (40, "test_fps_repeat_stress_500k"),
    "16k_brick_resampleILi16E": (32, "test_resample_k12_repeat_stress"),      # K + 1 in 10..13: not on the cycle
    "9k_brick_hILi2E": (1, "test_bandwidth_two_views_repeat_stress"),         # two views: not on the cycle
    "8k_rasterILi8ELb1ELb1E": (1, "test_forward_massive_depth_ties_bit_exact"),  # a pixel coordinate; tests/test_ties_gpu.py
}


def test_vgpr_spills_only_where_allowed_and_pinned():
    """No kernel of the library spills vector registers except the allow-list above, and no allow-listed kernel spills more
    than recorded; each entry names the -m gpu repeat-stress test that pins its results bit for bit at full size."""
    ks = _code_object_kernels()
    src = open(os.path.join(ROOT, "tests", "test_round6_gpu.py")).read() + open(os.path.join(ROOT, "tests", "test_ties_gpu.py")).read()
    seen = set()
    for name, vg, sg, priv in ks:
        if vg == 0:
            continue
        key = [k for k in SPILL_ALLOWED if k in name]
        assert key, "kernel %s spills %d VGPRs (%d B/lane of scratch) and is not on the allow-list" % (name, vg, priv)
        limit, test = SPILL_ALLOWED[key[0]]
        assert vg <= limit, "%s: %d spilled VGPRs, recorded ceiling %d" % (name, vg, limit)
        assert ("def %s(" % test) in src, test
        seen.add(key[0])
    # the cycle's own non-MFMA kernels, by name: zero
    for pat in ("16k_brick_resampleILi10E", "9k_brick_hILi4E", "13k_splat_front", "16k_splat_backward", "22k_splat_backward_heavy",
                "9k_bin_ldsILb", "11k_z_scatter", "13k_brick_count", "15k_brick_scatter", "16k_brick_offsets1"):
        hit = [k for k in ks if pat in k[0]]
        assert hit and all(k[1] == 0 for k in hit), (pat, hit)


def test_no_kernel_of_the_library_is_hit_by_the_structurizer_defect():
    """tools/structurize_scan.py: every source compiled to device IR with the Makefile's flags, taken through the code
    generator's IR passes in one process, and every multi-predecessor block that consists of zero-cost instructions only (the
    precondition of the defect: ~100 in the library, most of them extracts of vector loads) looked up again behind
    `structurizecfg` (with the code generator's predecessor orders, and with every block's predecessors reversed): none may have
    been emptied, i.e. hoisted into one of its predecessors -- that is the miscompile that
    gave round 5's raster build its wrong lists.  The detector's control is the reproducer: at plain -O3 it is hit."""
    import importlib.util
    import shutil
    import subprocess
    import pytest
    if not (shutil.which("/opt/rocm/bin/hipcc") and shutil.which("/opt/rocm/lib/llvm/bin/opt")):
        pytest.skip("no ROCm compiler / opt here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "structurize_scan.py")], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "hoisted out of a multi-predecessor block: 0" in r.stdout, r.stdout[-3000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "structurize_scan.py"), "--probe"], stdout=subprocess.PIPE, text=True)
    lines = c.stdout.strip().splitlines()
    assert len(lines) == 2 and lines[1].endswith("block: 0"), c.stdout            # with the library's flag: clean
    if c.returncode == 0:
        print("the reproducer is no longer hit at plain -O3: this toolchain does not show the defect")
