#!/bin/bash
# A/B of SIREN step-kernel variants on the GPU box: tools/ab_siren.sh [variant names under tools/variants] -> gpurun_out/ab_siren.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/ab_siren.txt
: > $OUT
for rep in 1 2; do
  python tools/siren_eval_bench.py 983040 >> $OUT 2>&1
  for v in "$@"; do
    ISO_DEV_LIB=tools/variants/libiso_$v.so python tools/siren_eval_bench.py 983040 >> $OUT 2>&1
  done
done
cat $OUT
