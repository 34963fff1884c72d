#!/bin/bash
# ISA-level variants of the failing build (haz_nospill): the device assembly hipcc wrote (-save-temps) is edited by
# asm_edit.py inside k_raster<8, true> only, re-assembled, linked, bundled and put behind the same host object -- the
# sub-commands are the ones `hipcc -v` prints (cc1as, lld, clang-offload-bundler, host cc1 with -fcuda-include-gpubinary).
#   tools/probes/spill_kit/make.sh && tools/probes/spill_kit/asm_variants.sh      -> tree/libiso_asm_{ctl,noswap,...}.so
#   modes: ctl (unedited: must fail), noswap (v_swap_b32 -> three v_xor), nomov64 (v_mov_b64 -> two v_mov_b32), nops (s_nop 7
#   behind every write of EXEC), cmpnops / vccnops (s_nop 7 behind every compare that writes an SGPR pair / VCC), waits
#   (s_waitcnt vmcnt(0) lgkmcnt(0) around every scratch access), initHEX / sinitHEX (all VGPRs / all SGPRs + VCC set at kernel
#   entry), vccz (VCCZ / EXECZ re-derived in front of every branch on them), delay (0.4 ms of s_sleep before the first list load)
set -e
REPO=$(cd "$(dirname "$0")/../../.." && pwd)
T=$REPO/tools/probes/spill_repro/tree
W=${W:-/tmp/asmx}
rm -rf $W && mkdir -p $W/base
cp $T/v_haz_nospill.hip $T/iso_points_amd/csrc/splat_asm.hip
( cd $W && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I$T/include \
    -save-temps -v -c $T/iso_points_amd/csrc/splat_asm.hip -o $W/splat_asm.o 2>&1 | grep -E '^ *"/opt/rocm' > $W/cmds.txt )
rm $T/iso_points_amd/csrc/splat_asm.hip
cp $W/splat_asm-hip-amdgcn-amd-amdhsa-gfx950.s $W/base/dev.s
python3 - "$W" <<'PY'
import sys
W = sys.argv[1]
cmds = open(W + "/cmds.txt").read().split("\n")
keep = [cmds[i].strip().replace(W + "/splat_asm.o", "splat_asm.o") for i in (3, 4, 5, 7, 8, 9)]   # device: as, lld, bundle; host: bc, S, as
open(W + "/replay.sh", "w").write("#!/bin/bash\nset -e\ncd $1\n" + "\n".join(keep) + "\n")
PY
chmod +x $W/replay.sh
for m in ${MODES:-ctl noswap nomov64 nops cmpnops vccnops waits}; do
  mkdir -p $W/v_$m && cp $W/splat_asm-host-x86_64-unknown-linux-gnu.hipi $W/v_$m/
  python3 $REPO/tools/probes/spill_kit/asm_edit.py $W/base/dev.s $W/v_$m/splat_asm-hip-amdgcn-amd-amdhsa-gfx950.s $m
  $W/replay.sh $W/v_$m > $W/v_$m/log.txt 2>&1
  OBJS=""; for o in $T/build/*.o; do b=$(basename $o); case $b in v_*) ;; splat.o) OBJS="$OBJS $W/v_$m/splat_asm.o";; *) OBJS="$OBJS $o";; esac; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/libiso_asm_$m.so $OBJS
done
ls $T/libiso_asm_*.so
