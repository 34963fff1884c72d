#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command + PMC passes for the dominant kernel;
# small text summaries land in gpurun_out/ (copy the ones to keep into profiles/).
# usage (on the GPU box, from the repo root): tools/profile_bench.sh TAG [steps]
TAG=${1:-v}; STEPS=${2:-3}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd $REPO && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1 )
tail -1 /tmp/prof_$TAG.log | cut -c1-400 > $REPO/gpurun_out/${TAG}_bench_under_rocprof.json
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB $REPO/gpurun_out/${TAG}_bench_kernel_stats.txt
head -30 $REPO/gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-70,86-150
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  ( cd $REPO && timeout 900 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_${TAG}_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_${TAG}_$i.log 2>&1 )
  DB=$(find /tmp/pmc_${TAG}_$i -name "*.db" | head -1)
  python $REPO/tools/pmc_summary.py $DB $REPO/gpurun_out/${TAG}_pmc_$i.txt
  head -8 $REPO/gpurun_out/${TAG}_pmc_$i.txt | cut -c1-60,90-170
done
