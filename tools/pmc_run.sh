#!/bin/bash
# Collect PMC counter sets (one rocprofv3 pass per set, --pmc with --kernel-trace only) for a command
# and write small per-kernel summaries under gpurun_out/.  usage: tools/pmc_run.sh TAG "cmd" "SET1" "SET2" ...
TAG=$1; CMD=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$TAG_$i
  ( cd $REPO && timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_${TAG}_$i -- $CMD > /tmp/pmc_${TAG}_$i.log 2>&1 )
  DB=$(find /tmp/pmc_${TAG}_$i -name "*.db" | head -1)
  mkdir -p $REPO/gpurun_out
  python $REPO/tools/pmc_summary.py $DB $REPO/gpurun_out/pmc_${TAG}_$i.txt
  head -14 $REPO/gpurun_out/pmc_${TAG}_$i.txt | cut -c1-60,90-170
done
