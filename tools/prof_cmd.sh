#!/bin/bash
# kernel-trace summary of an arbitrary command: tools/prof_cmd.sh TAG "cmd ..."  -> gpurun_out/TAG_kernel_stats.txt
TAG=$1; CMD=$2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd $REPO && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- $CMD > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB $REPO/gpurun_out/${TAG}_kernel_stats.txt
head -40 $REPO/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-70,87-150
