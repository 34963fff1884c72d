#!/bin/bash
# headline bench (short) for values of an environment knob: tools/r04_bench_ab.sh VAR v1 v2 ...
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  echo "== $VAR=$v"
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  siren share %.3f  launches %.1f  avg launch ms %.4f  frac %.4f  cfg3a %.4f ms' % (d['ms_per_step'], r['share_of_step'], r['launches_per_step'], r['avg_launch_ms'], r['frac'], d['cfg3a_analytic_sdf']['ms_per_step']))
print(r['active_points_per_launch_rank0'])"
done
