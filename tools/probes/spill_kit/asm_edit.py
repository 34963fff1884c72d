"""ISA-level edits of k_raster<8, true> in the device assembly of a spill_kit variant (asm_variants.sh).
usage: asm_edit.py IN.s OUT.s MODE"""
import re, sys
src, dst, mode = sys.argv[1], sys.argv[2], sys.argv[3]
L = open(src).read().split("\n")
# the function k_raster<8, true>
a = next(i for i, l in enumerate(L) if l.startswith("_ZN12_GLOBAL__N_18k_rasterILi8ELb1EEE") and l.rstrip().endswith(":") or re.match(r"^_ZN12_GLOBAL__N_18k_rasterILi8ELb1EEE\S*:\s", l))
b = next(i for i in range(a, len(L)) if L[i].startswith(".Lfunc_end") )
out = L[:a]
n = 0
for l in L[a:b]:
    s = l.strip()
    if mode == "noswap" and s.startswith("v_swap_b32"):
        m = re.match(r"v_swap_b32\s+(v\d+),\s*(v\d+)", s)
        x, y = m.group(1), m.group(2)
        out += ["\tv_xor_b32_e32 %s, %s, %s" % (x, x, y), "\tv_xor_b32_e32 %s, %s, %s" % (y, x, y), "\tv_xor_b32_e32 %s, %s, %s" % (x, x, y)]
        n += 1
        continue
    if mode == "nomov64" and s.startswith("v_mov_b64_e32"):
        m = re.match(r"v_mov_b64_e32\s+v\[(\d+):(\d+)\],\s*v\[(\d+):(\d+)\]", s)
        if m:
            d0, d1, s0, s1 = map(int, m.groups())
            if d0 == s1:      # low destination overlaps high source: high half first
                out += ["\tv_mov_b32_e32 v%d, v%d" % (d1, s1), "\tv_mov_b32_e32 v%d, v%d" % (d0, s0)]
            else:
                out += ["\tv_mov_b32_e32 v%d, v%d" % (d0, s0), "\tv_mov_b32_e32 v%d, v%d" % (d1, s1)]
            n += 1
            continue
    if mode == "vccz" and (s.startswith("s_cbranch_vccz") or s.startswith("s_cbranch_vccnz")):
        out += ["\ts_mov_b64 vcc, vcc", "\ts_nop 1"]; n += 1          # (the SI / CI work-around: re-derive VCCZ right in front of the branch)
    if mode == "vccz" and (s.startswith("s_cbranch_execz") or s.startswith("s_cbranch_execnz")):
        out += ["\ts_mov_b64 exec, exec", "\ts_nop 1"]; n += 1
    if mode == "sccnop" and s.startswith("s_cbranch_scc"):
        out += ["\ts_nop 1"]; n += 1
    if mode == "delay" and "global_load_dword" in s and " sc1" in s and not globals().get("_delayed"):
        out += ["\ts_sleep 127"] * 100           # ~0.4 ms in front of the first load of another slice's list (visibility lag?)
        globals()["_delayed"] = True; n += 1
    out.append(l)
    # blanket modes: behind EVERY instruction of the kernel an s_nop 7 (allnops: no wait-state hazard of any kind can be left),
    # an s_waitcnt vmcnt(0) lgkmcnt(0) (allwaits: no memory result is consumed early, no two memory operations overlap), or both
    if mode in ("allnops", "allwaits", "allboth") and s and not s.startswith((".", ";", "s_endpgm", "s_getpc", "s_setpc", "s_swappc")) and not s.endswith(":") \
            and not re.match(r"^[.\w$]+:", s) and "@rel32" not in s and "s_getpc" not in "".join(out[-3:]):
        if mode in ("allwaits", "allboth"):
            out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
        if mode in ("allnops", "allboth"):
            out.append("\ts_nop 7")
        n += 1
    if mode == "nops" and (s.startswith("s_or_b64 exec") or s.startswith("s_and_saveexec_b64") or s.startswith("s_andn2_saveexec_b64") or s.startswith("s_mov_b64 exec")):
        out.append("\ts_nop 7"); n += 1
    if mode == "cmpnops" and re.match(r"v_cmp\w+_e64\s+s\[", s):
        out.append("\ts_nop 7"); n += 1
    if mode == "waits" and s.startswith("scratch_"):
        out.insert(len(out) - 1, "\ts_waitcnt vmcnt(0) lgkmcnt(0)")
        out.append("\ts_waitcnt vmcnt(0)"); n += 1
    if mode == "vccnops" and (s.startswith("v_cmp") and "vcc" in s.split(",")[0]):
        out.append("\ts_nop 7"); n += 1
if mode.startswith("init"):           # every VGPR but v0 (the work-item ids) set at kernel entry: init0 / init7fffffff / ...
    val = int(mode[4:] or "0", 16)
    k = next(i for i in range(len(out)) if i > a and out[i].strip().startswith("; %bb.0"))
    out[k + 1:k + 1] = ["\tv_mov_b32_e32 v%d, 0x%x" % (r, val) for r in range(1, 76)]
    n = 75
if mode.startswith("sinit"):          # every SGPR but s[0:1] (kernarg pointer) and s2 (workgroup id), and VCC, set at kernel entry
    val = int(mode[5:] or "0", 16)
    k = next(i for i in range(len(out)) if i > a and out[i].strip().startswith("; %bb.0"))
    out[k + 1:k + 1] = ["\ts_mov_b32 s%d, 0x%x" % (r, val) for r in range(3, 100)] + ["\ts_mov_b32 vcc_lo, 0x%x" % val, "\ts_mov_b32 vcc_hi, 0x%x" % val]
    n = 99
out += L[b:]
open(dst, "w").write("\n".join(out))
print(mode, "edits:", n, "function lines", b - a)
