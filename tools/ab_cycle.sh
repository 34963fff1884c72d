#!/bin/bash
# per-kernel durations of the cfg-3a cycle for library variants: tools/ab_cycle.sh PATTERN [variant ...]  ("default" = the tree's library)
cd "$(dirname "$0")/.."
PAT=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=tools/variants/libiso_$v.so; fi
  echo "== $v"
  tools/seq_cmd.sh ab_$v 2>&1 | grep -E "cfg3a cycle|^# "
  grep -E "$PAT" gpurun_out/ab_${v}_kernel_stats.txt | cut -c1-50,87-140
done
