#!/bin/bash
# registers / spills / LDS of every kernel of one source file: tools/kres.sh siren_x3 ["-DFLAG ..."] [name-pattern]
cd "$(dirname "$0")/.."
EXTRA=$(grep -E "^FLAGS_$1 *:=" Makefile | sed 's/^[^=]*= *//')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Iinclude $EXTRA $2 \
  -Rpass-analysis=kernel-resource-usage -c ${KRES_DIR:-iso_points_amd/csrc}/$1.hip -o /tmp/kres_$$.o 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - - | \
  sed 's/  */ /g' | grep -E "${3:-.}"
rm -f /tmp/kres_$$.o
