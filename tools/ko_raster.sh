#!/bin/bash
cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for v in default KO_A KO_B hpt2; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=tools/variants/libiso_$v.so; fi
  tools/seq_cmd.sh ko_$v > /dev/null 2>&1
  echo "$v: $(grep -E 'k_raster<|k_raster_merge' gpurun_out/ko_${v}_sequence.txt | awk '{print $3}' | tr '\n' ' ')"
done
