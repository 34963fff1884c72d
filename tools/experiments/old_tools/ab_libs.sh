#!/bin/bash
# headline bench (short) for several builds of the library: tools/ab_libs.sh LIB1 LIB2 ...   ("main" = the in-tree build)
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" == "main" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=$lib; fi
  echo "== $lib"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  siren share %.3f  launches %.1f  avg launch ms %.4f  frac %.4f  cfg3a %.4f ms' % (d['ms_per_step'], r['share_of_step'], r['launches_per_step'], r['avg_launch_ms'], r['frac'], d['cfg3a_analytic_sdf']['ms_per_step']))"
done
