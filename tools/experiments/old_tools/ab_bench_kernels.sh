#!/bin/bash
# per-kernel times of the whole bench (SIREN cycle + cfg 3a) for library variants: tools/ab_bench_kernels.sh PATTERN [variant ...]
cd "$(dirname "$0")/.."
PAT=$1; shift
for v in "$@"; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=tools/variants/libiso_$v.so; fi
  echo "== $v"
  tools/prof_cmd.sh abk_$v "python bench.py --steps 5 --warmup 2 --no-cpu-baseline" 2>&1 | grep -E "$PAT"
done
