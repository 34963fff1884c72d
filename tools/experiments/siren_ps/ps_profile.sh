#!/bin/bash
# The headline bench and its PMC pass 1 with the point-stationary SIREN kernel enabled (ISO_SIREN_PS=1) -> gpurun_out/r04_ps_*
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISO_SIREN_PS=1
( cd $REPO && timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r04_ps_bench.json 2> $OUT/r04_ps_bench.err )
python -c "import json;d=json.load(open('$OUT/r04_ps_bench.json'));print('bench PS', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:60])"
rm -rf /tmp/rp_pspmc
( cd $REPO && ISO_BENCH_GRAPHS=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/rp_pspmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/rp_pspmc.log 2>&1 )
DB=$(find /tmp/rp_pspmc -name "*.db" | head -1)
python $REPO/tools/pmc_summary.py $DB $OUT/r04_ps_pmc_1.txt
head -8 $OUT/r04_ps_pmc_1.txt | cut -c1-200
rm -rf /tmp/rp_pspmc2
( cd $REPO && ISO_BENCH_GRAPHS=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/rp_pspmc2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/rp_pspmc2.log 2>&1 )
DB=$(find /tmp/rp_pspmc2 -name "*.db" | head -1)
python $REPO/tools/pmc_summary.py $DB $OUT/r04_ps_pmc_2.txt
head -8 $OUT/r04_ps_pmc_2.txt | cut -c1-200
