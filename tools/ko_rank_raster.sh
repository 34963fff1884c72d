#!/bin/bash
# k_raster knock-outs (RS_KO_A: no candidate phase, RS_KO_B: no insertions, RS_KO_EPI: no epilogue; results wrong) at the scale
# of ONE rank's band of an 8-rank run: tools/ko_rank_raster.sh [siren|sphere] -> gpurun_out/kor_<variant>_<sdf>_sequence.txt
REPO=$(cd "$(dirname "$0")/.." && pwd)
K=${1:-siren}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-default KO_A KO_B KO_EPI}; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=$REPO/tools/variants/libiso_$v.so; fi
  rm -rf /tmp/rp_kor_$v
  ( cd $REPO && ISO_WORLDS=8 ISO_TRACE_RANK=${RANK:-3} timeout 600 rocprofv3 --kernel-trace -d /tmp/rp_kor_$v -- python tools/rank_share_bench.py $K 1000000 1 > /tmp/rp_kor_$v.log 2>&1 )
  DB=$(find /tmp/rp_kor_$v -name "*.db" | head -1)
  python $REPO/tools/rank_sequence.py $DB $REPO/gpurun_out/kor_${v}_${K}_sequence.txt > /dev/null
  echo "$v $K: $(grep -E 'k_raster<|k_raster_merge|k_bin_lds' $REPO/gpurun_out/kor_${v}_${K}_sequence.txt | awk '{print $1}' | tr '\n' ' ')"
done
