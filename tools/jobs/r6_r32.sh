#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras --no-exact"
for v in base plainst base plainst base plainst; do
  if [ $v != base ]; then export BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_$v.so; else unset BEER_HIP_LIB; fi
  timeout 300 $C2 > /dev/null 2>&1
  python - <<P
import json
d=json.load(open('bench_detail.json'))
print('$v', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})
P
done
