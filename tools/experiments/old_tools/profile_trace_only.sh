#!/bin/bash
# kernel-trace summary of the bench command only (no PMC passes): tools/profile_trace_only.sh TAG [steps]
TAG=${1:-v}; STEPS=${2:-3}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd $REPO && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1 )
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB $REPO/gpurun_out/${TAG}_bench_kernel_stats.txt
grep -E "anonymous" $REPO/gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-64,87-150 | head -32
