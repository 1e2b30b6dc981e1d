#!/bin/bash
# A/B build of one kernel file: tools/ab_build.sh NAME FILE "-DFLAG=1 ..."
# -> build_ab/libbeer_hip_NAME.so (use with BEER_HIP_LIB=build_ab/libbeer_hip_NAME.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; DEFS=$3
# (per-file flags of beer_amd/build.py)
if { [ "$FILE" == "estep_bf16" ] || [ "$FILE" == "sample_grad" ]; } && [[ "$DEFS" != *-fslp-vectorize* ]]; then DEFS="$DEFS -fno-slp-vectorize"; fi
mkdir -p build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -mllvm -pragma-unroll-threshold=262144 \
  -Wno-unused-function -Iinclude -Ibeer_amd/csrc $DEFS -c beer_amd/csrc/$FILE.hip -o build_ab/${FILE}_$NAME.o
OBJS=""
for f in beer_amd/csrc/*.o; do
  b=$(basename $f .o)
  if [ "$b" == "$FILE" ]; then OBJS="$OBJS build_ab/${FILE}_$NAME.o"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libbeer_hip_$NAME.so $OBJS
echo build_ab/libbeer_hip_$NAME.so
