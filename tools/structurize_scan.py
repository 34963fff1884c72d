"""Does any kernel of the library hold the shape that this toolchain's StructurizeCFG miscompiles?

The defect (profiles/HISTORY.md, round 6; tools/probes/structurize_kit): a block with TWO OR MORE predecessors whose body is
nothing but zero-cost instructions (the insertelement / extractelement / shufflevector the SLP vectoriser makes of a swap; a
bitcast would do as well) is hoisted into its first predecessor and the values of the paths through the other predecessors
are lost.  This script compiles every source of the library to device IR with the flags of the Makefile, takes it through the
code generator's IR passes up to the one in front of `structurizecfg`, and lists the blocks of that shape per kernel:
  strict  all instructions of the block are vector inserts / extracts / shuffles / bitcasts   (the shape that was hit)
  broad   ... or any other instruction a target may price at zero (casts, freeze, fneg, address arithmetic)
and then looks at the same blocks BEHIND `structurizecfg` (same process, so with the predecessor orders the code generator really
has; blocks are named first with `instnamer` so that they can be found again) and behind `opt -passes=structurizecfg` run alone
on the IR in front of it with the predecessors of EVERY block reversed: a listed block that has become EMPTY in either was hoisted
out of a multi-predecessor position -- that is the miscompile itself, not merely its precondition.
usage: python tools/structurize_scan.py [-v] [file.hip ...]   (default: every file of iso_points_amd/csrc; ~15 s)
       python tools/structurize_scan.py --probe               the detector's own control: tools/probes/tie_merge.hip at plain -O3
                                                              (blocks are hoisted: exit 1) and with -fno-slp-vectorize (none)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLC = "/opt/rocm/lib/llvm/bin/llc"
OPT = "/opt/rocm/lib/llvm/bin/opt"
STRICT = {"insertelement", "extractelement", "shufflevector", "bitcast"}
BROAD = STRICT | {"addrspacecast", "freeze", "trunc", "zext", "sext", "fneg", "getelementptr", "inttoptr", "ptrtoint", "extractvalue", "insertvalue"}


def compile_lines():
    out = subprocess.run(["make", "-n", "-B", "-C", ROOT, "iso_points_amd/libisopoints_hip.so"], stdout=subprocess.PIPE, text=True).stdout
    return {os.path.basename([w for w in l.split() if w.endswith(".hip")][0]): l.split() for l in out.splitlines() if " -c " in l and ".hip" in l}


def scan_ir(ir):
    """{function: [(block, n_preds, opcodes)]} for blocks of the shape"""
    hits = {}
    for m in re.finditer(r"^define [^\n]*@([\w.$]+)\([^\n]*\{\n(.*?)^\}", ir, flags=re.M | re.S):
        fn, body = m.group(1), m.group(2)
        for b in re.finditer(r"^([\w.$]+):\s*; preds = ([^\n]*)\n(.*?)(?=^[\w.$]+:|\Z)", body, flags=re.M | re.S):
            preds = [p for p in b.group(2).split(",") if p.strip()]
            if len(preds) < 2:
                continue
            ops = []
            for line in b.group(3).splitlines():
                line = line.strip()
                if not line or line.startswith(";"):
                    continue
                mm = re.match(r"(?:%[\w.$]+ = )?(?:tail |musttail |notail )?([a-z_]+)", line)
                ops.append(mm.group(1) if mm else "?")
            core = [o for o in ops[:-1] if o != "phi"]                     # last one is the terminator
            if core and all(o in BROAD for o in core):
                hits.setdefault(fn, []).append((b.group(1), len(preds), core, all(o in STRICT for o in core)))
    return hits


def empty_blocks(ir):
    out = set()
    for m in re.finditer(r"^define [^\n]*@([\w.$]+)\([^\n]*\{\n(.*?)^\}", ir, flags=re.M | re.S):
        fn, body = m.group(1), m.group(2)
        for b in re.finditer(r"^([\w.$]+):[^\n]*\n(.*?)(?=^[\w.$]+:|\Z)", body, flags=re.M | re.S):
            lines = [l.strip() for l in b.group(2).splitlines() if l.strip() and not l.strip().startswith(";")]
            if len(lines) == 1 and lines[0].startswith("br label"):
                out.add((fn, b.group(1)))
    return out


def reversed_order(ir, tmp, base):
    """blocks emptied by `opt -passes=structurizecfg` when every multi-predecessor block's use list is reversed"""
    direct = []
    for m in re.finditer(r"^define [^\n]*@([\w.$]+)\([^\n]*\{\n(.*?)^\}", ir, flags=re.M | re.S):
        fn, body = m.group(1), m.group(2)
        for b in re.finditer(r"^([\w.$]+):\s*; preds = ([^\n]*)\n", body, flags=re.M):
            n = len([p for p in b.group(2).split(",") if p.strip()])
            if n >= 2 and not b.group(1)[0].isdigit():
                direct.append((fn, b.group(1), n))
    skip = set()
    for _ in range(40):
        txt = ir + "\n" + "".join("uselistorder_bb @%s, %%%s, { %s }\n" % (fn, blk, ", ".join(str(i) for i in range(n - 1, -1, -1)))
                                  for fn, blk, n in direct if (fn, blk) not in skip)
        f = os.path.join(tmp, base + ".rev.ll")
        open(f, "w").write(txt)
        r = subprocess.run([OPT, "-S", "-passes=structurizecfg", f, "-o", f + ".out"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode == 0:
            return empty_blocks(open(f + ".out").read())
        m = re.search(r":(\d+):\d+: error", r.stderr)                 # a directive whose count does not fit (duplicate edges): drop it
        if not m:
            raise RuntimeError(r.stderr[:500])
        bad = txt.splitlines()[int(m.group(1)) - 1]
        mm = re.match(r"uselistorder_bb @([\w.$]+), %([\w.$]+),", bad)
        if not mm:
            raise RuntimeError(r.stderr[:500])
        skip.add((mm.group(1), mm.group(2)))
    raise RuntimeError("too many directives rejected")


def one(src, words, tmp):
    base = os.path.basename(src)[:-4]
    bc = os.path.join(tmp, base + ".bc")
    cmd = [w for w in words if w not in ("-c",)]
    o = cmd.index("-o")
    cmd = cmd[:o] + cmd[o + 2:]
    cmd = [c if not c.endswith(".hip") or os.path.isabs(c) else os.path.join(ROOT, c) for c in cmd]
    cmd = [c.replace("-Iinclude", "-I" + os.path.join(ROOT, "include")) for c in cmd]
    subprocess.check_call(cmd + ["-w", "--cuda-device-only", "-emit-llvm", "-c", "-o", bc], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    named = os.path.join(tmp, base + ".named.bc")
    subprocess.check_call([OPT, "-passes=instnamer", bc, "-o", named])

    def stage(stop):
        mir = os.path.join(tmp, base + "." + stop + ".mir")
        subprocess.check_call([LLC, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-stop-after=" + stop, named, "-o", mir],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t = open(mir).read()
        a = t.index("--- |") + 5
        b = t.index("\n...", a)
        return "\n".join(l[2:] if l.startswith("  ") else l for l in t[a:b].split("\n"))

    ir = stage("unify-loop-exits")
    hits = scan_ir(ir)
    post = stage("structurizecfg")
    emptied = set()
    for m in re.finditer(r"^define [^\n]*@([\w.$]+)\([^\n]*\{\n(.*?)^\}", post, flags=re.M | re.S):
        fn, body = m.group(1), m.group(2)
        for b in re.finditer(r"^([\w.$]+):[^\n]*\n(.*?)(?=^[\w.$]+:|\Z)", body, flags=re.M | re.S):
            lines = [l.strip() for l in b.group(2).splitlines() if l.strip() and not l.strip().startswith(";")]
            if len(lines) == 1 and lines[0].startswith("br label"):
                emptied.add((fn, b.group(1)))
    # ... and, because the defect depends on the ORDER of a block's predecessors (which a later edit of the source or another
    # compiler version may change), the same pass alone on the IR in front of it with the predecessors of every block REVERSED
    emptied |= reversed_order(ir, tmp, base)
    for fn, v in hits.items():
        hits[fn] = [h + ((fn, h[0]) in emptied,) for h in v]
    return base, "-fno-slp-vectorize" in words, len(re.findall(r"^define ", ir, flags=re.M)), hits


def main():
    if "--probe" in sys.argv:
        src = "tools/probes/tie_merge.hip"
        base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src, "-o", "x.o"]
        tot = 0
        with tempfile.TemporaryDirectory() as tmp:
            for words in (base, base + ["-fno-slp-vectorize"]):
                _, noslp, nfn, hits = one(src, words, tmp)
                nh = sum(1 for v in hits.values() for h in v if h[4])
                print("tie_merge.hip %-22s %d kernels: %d blocks of the strict shape, hoisted out of a multi-predecessor block: %d"
                      % ("(-fno-slp-vectorize)" if noslp else "(plain -O3)", nfn, sum(1 for v in hits.values() for h in v if h[3]), nh))
                tot += nh if not noslp else 0
        return 1 if tot else 0
    lines = compile_lines()
    want = [os.path.basename(a) for a in sys.argv[1:] if a.endswith(".hip")] or sorted(lines)
    strict = broad = hoisted = 0
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(8) as ex:
        for base, noslp, nfn, hits in ex.map(lambda f: one(os.path.join("iso_points_amd/csrc", f), lines[f], tmp), want):
            ns = sum(1 for v in hits.values() for h in v if h[3])
            nb = sum(len(v) for v in hits.values()) - ns
            nh = sum(1 for v in hits.values() for h in v if h[4])
            strict += ns
            broad += nb
            hoisted += nh
            print("%-18s %-22s %3d kernels / functions: %3d blocks of the strict shape, %3d more of the broad one; EMPTIED by structurizecfg: %d"
                  % (base + ".hip", "(built without SLP)" if noslp else "(built WITH SLP)", nfn, ns, nb, nh))
            for fn, v in sorted(hits.items()):
                for blk, np_, core, st, hz in v:
                    if hz or "-v" in sys.argv:
                        print("      %s %s  %s: block %s, %d predecessors: %s" % ("HOISTED" if hz else "       ", "strict" if st else "broad ", fn[:70], blk, np_, " ".join(core)))
    print("total: %d blocks of the strict shape, %d of the broad one; hoisted out of a multi-predecessor block: %d" % (strict, broad, hoisted))
    return 1 if hoisted else 0


if __name__ == "__main__":
    sys.exit(main())
