"""VGPR spills of every kernel in iso_points_amd/libisopoints_hip.so, read from the code objects' metadata
(.vgpr_spill_count of the amdhsa.kernels notes).  Spills are a performance matter here, not a correctness one: round 5
blamed two spilled VGPRs of a k_raster build for wrong lists on depth ties, round 6 showed them innocent (the defect is the
toolchain's StructurizeCFG on SLP-vectorised swap chains: profiles/HISTORY.md, tools/probes/structurize_kit).  The library
keeps an allow-list of spilling kernels with pinned ceilings (tests/test_abi.py) so that a rebuild that starts to spill gets
a look.
usage: python tools/spill_check.py [lib.so]   -> one line per spilling kernel, exit code 1 if any"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(lib):
    """[(kernel name, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size)] of all code objects in lib"""
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "x.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, a in enumerate(starts):
            b = starts[k + 1] if k + 1 < len(starts) else len(blob)
            part = os.path.join(d, "b%d.bin" % k)
            open(part, "wb").write(blob[a:b])
            co = os.path.join(d, "b%d.co" % k)
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(name|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s*(\S+)", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2)
                if key == "name" and not line.lstrip().startswith("- .name") and "cur_args" in cur:
                    pass
                cur[key] = val
                if key == "vgpr_spill_count":           # the last of a kernel's fields we need (alphabetical order of the notes)
                    if "name" in cur:
                        out.append((cur.get("name"), int(val), int(cur.get("sgpr_spill_count", 0)),
                                    int(cur.get("private_segment_fixed_size", 0))))
                    cur = {}
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "iso_points_amd", "libisopoints_hip.so")
    ks = kernels(lib)
    bad = [k for k in ks if k[1] > 0]
    for name, v, s, p in bad:
        print("%-110s VGPR spills %d (SGPR spills %d, scratch %d B/lane)" % (name[:110], v, s, p))
    print("%d kernels, %d with VGPR spills" % (len(ks), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
