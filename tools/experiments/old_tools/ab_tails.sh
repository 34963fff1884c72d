#!/bin/bash
# the two tail kernels in the SIREN cycle (bench trace) for library variants: tools/ab_tails.sh default wi16 ...
cd "$(dirname "$0")/.."
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" == "default" ]; then unset ISO_DEV_LIB; else export ISO_DEV_LIB=$R/tools/variants/libiso_$v.so; fi
  rm -rf /tmp/tl; ( cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/tl.log 2>&1 )
  python $R/tools/rocprof_summary.py $(find /tmp/tl -name "*.db" | head -1) /tmp/tl.txt > /dev/null
  echo "== $v"; grep -E "_tail" /tmp/tl.txt | cut -c1-60,87-140
done
