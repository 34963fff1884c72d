#!/bin/bash
# cfg-3a cycle per-kernel times for a sweep of the two grids' cell sizes: tools/sweep_cells.sh
cd "$(dirname "$0")/.."
for rc in 0.6 0.7 0.8 0.9; do
  export ISO_RESAMPLE_CELL=$rc; unset ISO_H_CELL_SCALE
  echo "== resample cell $rc r"
  tools/seq_cmd.sh sw 2>&1 | grep -E "^# "
  grep -E "k_brick_resample" gpurun_out/sw_kernel_stats.txt | cut -c1-50,87-130
done
unset ISO_RESAMPLE_CELL
for hc in 4.0 4.5 5.0 5.5 6.0 7.0; do
  export ISO_H_CELL_SCALE=$hc
  echo "== h cell $hc spacings"
  tools/seq_cmd.sh sw 2>&1 | grep -E "^# "
  grep -E "k_brick_h" gpurun_out/sw_kernel_stats.txt | cut -c1-50,87-130
done
