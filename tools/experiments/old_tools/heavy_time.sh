#!/bin/bash
# k_splat_backward_heavy in the SIREN cycle (denser gradient image than cfg 3a) and in the cfg-3a cycle
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp; R=$OLDPWD
rm -rf /tmp/hv; ( cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/hv.log 2>&1 )
python $R/tools/rocprof_summary.py $(find /tmp/hv -name "*.db" | head -1) /tmp/hv.txt > /dev/null
grep -E "k_splat_backward" /tmp/hv.txt | cut -c1-60,87-140
