#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras --no-exact --no-check"
for v in base r32 base r32 base r32; do
  if [ $v == r32 ]; then export BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_r32.so; else unset BEER_HIP_LIB; fi
  timeout 300 $C2 > /dev/null 2>&1
  python - <<P
import json
d=json.load(open('bench_detail.json'))
print('$v', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})
P
done
