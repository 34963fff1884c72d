#!/bin/bash
# Pins the compiler defect behind round 5's "wrong K-best lists on depth ties" ON THE HOST CPU, without a GPU:
# the merge loop of tools/probes/tie_merge.hip (one instantiation) is taken through the device pipeline of this toolchain
# stage by stage; the LLVM IR of every stage is retargeted to x86 (tox86.py: address spaces, the work-item / kernel-argument
# intrinsics and the calling convention replaced -- nothing else) and EXECUTED for 1024 lanes of lists full of equal depths
# beside the IR the optimiser was given (host_check.c).  Usage: tools/probes/structurize_kit/pin.sh  (about a minute;
# writes into /tmp/structurize_kit).  Result on ROCm 7.2.0 (AMD clang 22.0.0git roc-7.2.0 7b800a19): profiles/r06_tie_miscompile_pinned.txt
set -e
KIT=$(cd "$(dirname "$0")" && pwd); REPO=$(cd $KIT/../../.. && pwd)
W=/tmp/structurize_kit; rm -rf $W; mkdir -p $W; cd $W
BIN=/opt/rocm/lib/llvm/bin; LLC="$BIN/llc -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3"; OPT=$BIN/opt; CL=$BIN/clang
$CL --version | head -1
# one kernel: atomic loads, the first four entries of every list, K at run time (73 % of the lanes wrong on the GPU)
python3 - "$REPO" <<'PY'
import sys
s = open(sys.argv[1] + '/tools/probes/tie_merge.hip').read()
k = s[:s.index("struct E {")] + '\ntemplate __global__ void k_merge<0, true, 1, false>(const float*, int, int, int, int, int*, float*);\n'
open('km.hip', 'w').write(k)
PY
FLAGS="-w --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -emit-llvm km.hip"
# -opt-bisect-limit: 750 = every IR pass before the SLP vectoriser, 751 = ... and the SLP vectoriser (the pass numbers of this file)
/opt/rocm/bin/hipcc $FLAGS -o all.ll -mllvm -opt-bisect-limit=-1 2> bisect.txt || true
N=$(grep "slp-vectorizer" bisect.txt | head -1 | sed 's/.*(\([0-9]*\)).*/\1/'); echo "the SLP vectoriser is optional pass $N of $(grep -c 'BISECT: running' bisect.txt)"
/opt/rocm/bin/hipcc $FLAGS -o before_slp.ll -mllvm -opt-bisect-limit=$((N-1)) 2>/dev/null
/opt/rocm/bin/hipcc $FLAGS -o after_slp.ll -mllvm -opt-bisect-limit=$N 2>/dev/null
ir_of() { python3 - "$1" "$2" <<'PY'
import sys
t = open(sys.argv[1]).read(); a = t.index('--- |') + 5; b = t.index('\n...', a)
open(sys.argv[2], 'w').write('\n'.join(l[2:] if l.startswith('  ') else l for l in t[a:b].split('\n')))
PY
}
python3 $KIT/tox86.py before_slp.ll ref_x86.ll k_before && $CL -w -O0 -c ref_x86.ll -o ref.o
check() {  # file.ll, label
  python3 $KIT/tox86.py $1 t_x86.ll k_test && $CL -w -O0 -c t_x86.ll -o t.o && $CL -w -O1 $KIT/host_check.c ref.o t.o -o chk && ./chk "$2"
}
echo; echo "== executed on the host: lanes (of 1024) whose ids or q values differ from the IR before the SLP vectoriser"
check after_slp.ll "IR right after the SLP vectoriser"
for P in amdgpu-codegenprepare codegenprepare flattencfg sink amdgpu-late-codegenprepare amdgpu-unify-divergent-exit-nodes fix-irreducible unify-loop-exits structurizecfg; do
  $LLC -stop-after=$P after_slp.ll -o s.mir 2>/dev/null; ir_of s.mir s_$P.ll; check s_$P.ll "SLP, llc ... -> $P (in one process)"
done
$LLC -stop-after=structurizecfg before_slp.ll -o s.mir 2>/dev/null; ir_of s.mir n_structurizecfg.ll; check n_structurizecfg.ll "no SLP, llc ... -> structurizecfg"
echo; echo "== the IR in front of structurizecfg, written out, read back (that resets every block's predecessor order to the order of the text), then opt -passes=structurizecfg alone"
$OPT -S -passes=structurizecfg s_unify-loop-exits.ll -o a_good.ll; check a_good.ll "SLP, predecessors as parsed"
$LLC -stop-after=codegenprepare after_slp.ll -o c.mir 2>/dev/null; ir_of c.mir c.ll; $LLC -start-after=codegenprepare -stop-after=unify-loop-exits c.ll -o d.mir 2>/dev/null; ir_of d.mir d.ll
diff <(grep "; preds = " s_unify-loop-exits.ll) <(grep "; preds = " d.ll) | grep "^<" | sed 's/^< /   order in the one-process pipeline: /' || true
BL=$(diff <(grep "; preds = " s_unify-loop-exits.ll) <(grep "; preds = " d.ll) | grep "^<" | sed 's/^< \([0-9]*\):.*/\1/' | paste -sd,)
python3 $KIT/reverse_preds.py s_unify-loop-exits.ll rev.ll $BL > /dev/null; $OPT -S -passes=structurizecfg rev.ll -o bad4.ll; check bad4.ll "SLP, those blocks reversed (uselistorder_bb)"
python3 $KIT/reverse_preds.py s_unify-loop-exits.ll rev.ll all > /dev/null; $OPT -S -passes=structurizecfg rev.ll -o b.ll; check b.ll "SLP, all two-predecessor blocks reversed"
$LLC -stop-after=unify-loop-exits before_slp.ll -o s.mir 2>/dev/null; ir_of s.mir n_pre.ll
$OPT -S -passes=structurizecfg n_pre.ll -o a.ll; check a.ll "no SLP, predecessors as parsed"
python3 $KIT/reverse_preds.py n_pre.ll rev.ll all > /dev/null; $OPT -S -passes=structurizecfg rev.ll -o b.ll; check b.ll "no SLP, all two-predecessor blocks reversed"
echo; echo "== what structurizecfg made of the first of those blocks (the swap of level 0: reached from the \`cz < z[0]\` test AND from the tie test)"
echo "-- in front of the pass:"; B=$(echo $BL | cut -d, -f1); P=$(grep "^$B:" s_unify-loop-exits.ll | sed 's/.*preds = %\([0-9]*\),.*/\1/')
awk -v p="$P:" -v b="$B:" '$1==p{on=1} on{print "   " $0} on && /^$/ && seen{exit} $1==b{seen=1}' s_unify-loop-exits.ll | head -40
echo "-- behind it, predecessors in the one-process order: the swap block's two instructions sit in the first test's block, the block is empty, and a"
echo "   lane that arrives through the tie test takes the NO-SWAP values whatever the tie test said (the i1 phi still sends it through the empty block):"
L=$(grep -n "^bb$B:" bad4.ll | cut -d: -f1); sed -n "$((L-19)),$((L+1))p" bad4.ll | sed 's/^/   /'
echo "-- behind it, predecessors as parsed (correct):"
L=$(grep -n "^Flow97:" a_good.ll | cut -d: -f1); sed -n "$((L-12)),$((L+8))p" a_good.ll | sed 's/^/   /'
