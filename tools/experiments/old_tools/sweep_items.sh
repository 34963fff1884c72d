#!/bin/bash
# raster work-item target sweep: per-rank share at N = 8 (sphere cycle, graphs) and the one-GPU cfg-3a cycle
cd "$(dirname "$0")/.."
for t in 256 512 1024 2048 4096; do
  export ISO_RASTER_ITEMS=$t
  echo "== items $t"
  ISO_WORLDS=1,8 python tools/rank_share_bench.py sphere 1000000 5 graphs 2>&1 >/dev/null | grep -E "^[18] "
done
