#!/bin/bash
# A/B of SIREN step-kernel variants on a Newton projection (move epilogue included), three alternating rounds:
# tools/ab_project.sh T variant [variant ...]  -> gpurun_out/ab_project.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/ab_project.txt
T=$1; shift
: > $OUT
for rep in 1 2 3; do
  python tools/siren_step_time.py $T 2>/dev/null | grep median >> $OUT
  for v in "$@"; do
    ISO_DEV_LIB=tools/variants/libiso_$v.so python tools/siren_step_time.py $T 2>/dev/null | grep median >> $OUT
  done
done
cat $OUT
