#!/bin/bash
# Rebuilds round 5's "k_raster with two spilled VGPRs returns stale ids on depth ties" case and the variants that take it
# apart (round 6).  Everything is derived from the history of this repository: the tree at e06bcf9 (the state the archived
# experiment tools/experiments/raster_fold_merge.diff was cut from) is unpacked under tools/probes/spill_repro/tree
# (git-ignored), the diff is applied, and the variants below are single-statement edits of it.
#   tools/probes/spill_kit/make.sh            (here, ~4 min: one library per variant)
#   gpurun -- 'cd tools/probes/spill_repro/tree && for v in default fold haz haz_nospill haz_top haz_bot p1_wait0 p2_nop p3_plain p4_ownfirst haz_dbg haz_post; do python repro.py $v; done'
# Results: profiles/r06_spill_repro_[1-3].txt, DESIGN.md 3.3.
set -e
REPO=$(cd "$(dirname "$0")/../../.." && pwd)
T=$REPO/tools/probes/spill_repro/tree
rm -rf $T && mkdir -p $T
git -C $REPO archive e06bcf9 iso_points_amd include oracle Makefile tests/splat_util.py tests/util.py | tar -x -C $T
cp $REPO/tools/probes/spill_kit/repro.py $T/
( cd $T && make -j16 iso_points_amd/libisopoints_hip.so > /dev/null && make -C oracle > /dev/null )
cp $T/iso_points_amd/csrc/splat.hip $T/v_fold.hip
patch -s $T/v_fold.hip $REPO/tools/experiments/raster_fold_merge.diff
python3 $REPO/tools/probes/spill_kit/variants.py $T
HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I$T/include"
for f in $T/v_*.hip; do
  v=$(basename $f .hip); v=${v#v_}
  ( cp $f $T/iso_points_amd/csrc/splat_$v.hip
    /opt/rocm/bin/hipcc $HIPFLAGS -Rpass-analysis=kernel-resource-usage -c $T/iso_points_amd/csrc/splat_$v.hip -o $T/build/v_$v.o 2>&1 |
      grep -E "Function Name|VGPRs:|Spill|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - | grep "k_rasterILi8ELb1" | sed "s/^/$v: /" | cut -c1-24,170-300
    rm $T/iso_points_amd/csrc/splat_$v.hip
    OBJS=""; for o in $T/build/*.o; do b=$(basename $o); case $b in v_*) ;; splat.o) OBJS="$OBJS $T/build/v_$v.o";; *) OBJS="$OBJS $o";; esac; done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $T/libiso_$v.so $OBJS ) &
done
wait
ls $T/*.so
