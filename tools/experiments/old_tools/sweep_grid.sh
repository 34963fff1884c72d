#!/bin/bash
# grid sizes of the two fused brick kernels: tools/sweep_grid.sh ENVNAME PATTERN size [size ...]
cd "$(dirname "$0")/.."
NAME=$1; PAT=$2; shift; shift
for g in "$@"; do
  export $NAME=$g
  echo "== $NAME=$g"
  tools/seq_cmd.sh grid_$g 2>&1 | grep -E "^# "
  grep -E "$PAT" gpurun_out/grid_${g}_kernel_stats.txt | cut -c1-50,87-140
done
