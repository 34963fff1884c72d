#!/bin/bash
# per-kernel durations of an arbitrary command: tools/kstats.sh "python tools/stage_times.py" [grep-pattern]
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
( cd $REPO && timeout 600 rocprofv3 --kernel-trace -d /tmp/ks -- $1 > /tmp/ks.log 2>&1 )
DB=$(find /tmp/ks -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB /tmp/ks.txt
grep -E "${2:-anonymous}" /tmp/ks.txt | cut -c1-64,87-140 | head -${3:-30}
