#!/bin/bash
# k_splat_backward_heavy of one rank at N = 8 (SIREN cycle) for library variants: tools/ab_heavy_ranks.sh default ry16 ...
cd "$(dirname "$0")/.."
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=$R/tools/variants/libiso_$v.so; fi
  rm -rf /tmp/hr; ( cd $R && ISO_WORLDS=8 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/hr -- python tools/rank_share_bench.py siren 1000000 2 > /tmp/hr.log 2>&1 )
  python $R/tools/rocprof_summary.py $(find /tmp/hr -name "*.db" | head -1) /tmp/hr.txt > /dev/null
  echo "== $v"; grep -E "k_splat_backward_heavy" /tmp/hr.txt | cut -c1-60,87-140
done
