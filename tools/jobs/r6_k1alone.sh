#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in base r32 base r32; do
  if [ $v != base ]; then export BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_$v.so; else unset BEER_HIP_LIB; fi
  echo -n "$v: "; timeout 300 python tools/probes/k1_ab.py 2>/dev/null | grep "k1_lds=1" | tr '\n' ' '; echo
done
