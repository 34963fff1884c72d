#!/bin/bash
# Everything profiles/r06_* is made of (round 6; tools/profile_r05.sh is last round's), in one gpurun call (run from the repo root on the GPU box):
#   tools/profile_r06.sh [part ...]      parts: bench trace pmc cfg3a ranks micro configs power idr opapi  (default: the first seven)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PARTS=${@:-"bench trace pmc cfg3a ranks micro configs"}
has() { [[ " $PARTS " == *" $1 "* ]]; }
run_prof() {  # tag, extra rocprof args, command...
  local TAG=$1; shift; local ARGS=$1; shift
  rm -rf /tmp/rp_$TAG
  ( cd $REPO && timeout 1200 rocprofv3 --kernel-trace $ARGS -d /tmp/rp_$TAG -- "$@" > /tmp/rp_$TAG.log 2>&1 )
  find /tmp/rp_$TAG -name "*.db" | head -1
}
if has bench; then
  ( cd $REPO && timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/r06_bench.json 2> $OUT/r06_bench.err )
  python -c "import json;d=json.load(open('$OUT/r06_bench.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cfg3a_analytic_sdf']['ms_per_step'], d['f32_mfma_mode']['ms_per_step'], d['cpu_baseline']['value'])"
fi
if has trace; then
  DB=$(run_prof bench "--stats" python bench.py --steps 5 --warmup 2 --no-cpu-baseline)
  python $REPO/tools/rocprof_summary.py $DB $OUT/r06_bench_kernel_stats.txt
  head -12 $OUT/r06_bench_kernel_stats.txt | cut -c1-70,87-150
fi
if has pmc; then
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DB=$(ISO_BENCH_GRAPHS=0 run_prof pmc$i "--pmc $SET" python bench.py --steps 2 --warmup 1 --no-cpu-baseline)
    python $REPO/tools/pmc_summary.py $DB $OUT/r06_pmc_$i.txt
  done
  python $REPO/tools/traffic_from_pmc.py $OUT/r06_pmc_3.txt $OUT/r06_pmc_4.txt $OUT/r06_traffic.json
  cat $OUT/r06_traffic.json
fi
if has cfg3a; then
  DB=$(run_prof cfg3a "--stats" python tools/cycle_only.py 10)
  python $REPO/tools/rocprof_summary.py $DB $OUT/r06_cfg3a_kernel_stats.txt
  grep "cfg3a cycle" /tmp/rp_cfg3a.log
  python $REPO/tools/cycle_sequence.py $DB $OUT/r06_cfg3a_sequence.txt | tail -1
  i=0
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DB=$(run_prof c3a$i "--pmc $SET" python tools/cycle_only.py 4)
    python $REPO/tools/pmc_summary.py $DB $OUT/r06_cfg3a_pmc_$i.txt
  done
fi
if has ranks; then
  ( cd $REPO && timeout 1500 python tools/rank_share_bench.py siren 1000000 5 > $OUT/r06_rank_share_siren.json 2> $OUT/r06_rank_share_siren.err; tail -4 $OUT/r06_rank_share_siren.err
    timeout 900 python tools/rank_share_bench.py sphere 1000000 5 graphs > $OUT/r06_rank_share_sphere.json 2> $OUT/r06_rank_share_sphere.err; tail -4 $OUT/r06_rank_share_sphere.err )
  DB=$(ISO_WORLDS=8 run_prof rs8 "--stats" python tools/rank_share_bench.py siren 1000000 4)
  python $REPO/tools/rocprof_summary.py $DB $OUT/r06_rank_share_world8_kernel_stats.txt
fi
if has micro; then
  ( cd $REPO && timeout 300 python tools/bricks_bench.py > $OUT/r06_bricks_bench.json 2>/dev/null )
fi
if has configs; then
  ( cd $REPO && timeout 900 python tools/configs_bench.py > $OUT/r06_configs_bench.log 2>&1; cp $OUT/configs_bench.json $OUT/r06_configs_bench.json; tail -12 $OUT/r06_configs_bench.log | cut -c1-200 )
fi
if has power; then
  # verdict r4 item 4: socket power + gfx clock (amdsmi, ~50 Hz) during >= 3 s of each kernel
  ( cd $REPO && timeout 600 python tools/power_probe.py $OUT/r06_power_siren.txt \
      "siren_x3_both=python tools/siren_loop.py 4" \
      "siren_ps=ISO_DEV_LIB=tools/variants/libiso_siren_ps.so ISO_SIREN_PS=1 python tools/siren_loop.py 4" \
      "siren_f32_mfma=ISO_SIREN_GEMM=f32 python tools/siren_loop.py 4" \
      "mfma_zeros=tools/probes/mfma_power long 3 4" \
      "mfma_smooth=tools/probes/mfma_power long 1 4" \
      "mfma_random=tools/probes/mfma_power long 2 4" \
      "cfg3a_cycle=python tools/cycle_only.py 3000" > $OUT/r06_power.log 2>&1; tail -60 $OUT/r06_power.log | cut -c1-200 )
fi
if has idr; then
  # verdict r4 item 5: configs[3]'s dominant kernel (IDR 8x512 projection), kernel stats + PMC passes 1-4
  DB=$(run_prof idr "--stats" python tools/idr_bench.py 1000000)
  python $REPO/tools/rocprof_summary.py $DB $OUT/r06_idr_kernel_stats.txt
  grep IDR /tmp/rp_idr.log
  head -8 $OUT/r06_idr_kernel_stats.txt | cut -c1-70,87-150
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DB=$(run_prof idr$i "--pmc $SET" python tools/idr_one.py)
    python $REPO/tools/pmc_summary.py $DB $OUT/r06_idr_pmc_$i.txt
    head -6 $OUT/r06_idr_pmc_$i.txt | cut -c1-200
  done
fi
if has opapi; then
  TAGO=${OPAPI_TAG:-r06_opapi}
  DB=$(run_prof opapi "" python tools/opapi_only.py 6 trace)
  grep -i "operator API" /tmp/rp_opapi.log | cut -c1-200
  python $REPO/tools/opapi_sequence.py $DB $OUT/${TAGO}_sequence.txt | tail -1
fi
if has rankseq; then
  # what ONE rank of an 8-rank run launches per cycle (markers around rank 3's segments)
  for K in siren sphere; do
    DB=$(ISO_WORLDS=8 ISO_TRACE_RANK=3 run_prof rseq_$K "" python tools/rank_share_bench.py $K 1000000 1)
    python $REPO/tools/rank_sequence.py $DB $OUT/${RANKSEQ_TAG:-r06}_rank3_of_8_${K}_sequence.txt | tail -1
  done
fi
