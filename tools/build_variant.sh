#!/bin/bash
# Development aid: build a variant of libisopoints_hip.so with extra flags for ONE source file
# (timing experiments; see the X3_DBG_* hooks in csrc/siren_x3.hip).
# usage: tools/build_variant.sh NAME FILE.hip "-DFLAG ..."   -> tools/variants/libiso_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p tools/variants build/var_$NAME
make -s iso_points_amd/libisopoints_hip.so >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Iinclude $FLAGS \
  -c iso_points_amd/csrc/$SRC -o build/var_$NAME/${SRC%.hip}.o
OBJS=""
for o in build/*.o; do
  b=$(basename $o)
  if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS build/var_$NAME/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/variants/libiso_$NAME.so $OBJS
echo tools/variants/libiso_$NAME.so
