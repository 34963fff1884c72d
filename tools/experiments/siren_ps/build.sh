#!/bin/bash
# The point-stationary SIREN step (siren_ps.hip; bit-identical to the shipped k_siren_step_x3, same speed: profiles/HISTORY.md
# rounds 4-5) as a VARIANT library: tools/variants/libiso_siren_ps.so = the product objects with siren_x3.hip compiled
# -DISO_WITH_SIREN_PS + siren_ps.o.  Select it with ISO_DEV_LIB=tools/variants/libiso_siren_ps.so ISO_SIREN_PS=1.
set -e
cd "$(dirname "$0")/../../.."
make -s iso_points_amd/libisopoints_hip.so >/dev/null
mkdir -p tools/variants build/var_siren_ps
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Iinclude -Iiso_points_amd/csrc -DISO_WITH_SIREN_PS"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=max-memory-clause -c iso_points_amd/csrc/siren_x3.hip -o build/var_siren_ps/siren_x3.o
/opt/rocm/bin/hipcc $F -c tools/experiments/siren_ps/siren_ps.hip -o build/var_siren_ps/siren_ps.o
OBJS=$(ls build/*.o | grep -v "build/siren_x3.o" | grep -v "build/siren_ps.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/libiso_siren_ps.so $OBJS build/var_siren_ps/siren_x3.o build/var_siren_ps/siren_ps.o
echo tools/variants/libiso_siren_ps.so
