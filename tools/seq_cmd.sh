#!/bin/bash
# ordered kernel sequence of one cfg-3a cycle: tools/seq_cmd.sh TAG -> gpurun_out/TAG_sequence.txt (+ TAG_kernel_stats.txt)
TAG=$1
MODE=$2   # "siren": the headline cycle
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd $REPO && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -- python tools/cycle_only.py 10 $MODE > /tmp/prof_$TAG.log 2>&1 )
tail -2 /tmp/prof_$TAG.log
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB $REPO/gpurun_out/${TAG}_kernel_stats.txt
python $REPO/tools/cycle_sequence.py $DB $REPO/gpurun_out/${TAG}_sequence.txt
