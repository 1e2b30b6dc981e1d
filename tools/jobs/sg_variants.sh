#!/bin/bash
# time of the sample-gradient kernel with A/B libraries: name ...
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=build_ab/libbeer_hip_$v.so; fi
  echo -n "$v  "; timeout 150 python tools/probes/sgrad_time.py 2>&1 | tail -1
done
