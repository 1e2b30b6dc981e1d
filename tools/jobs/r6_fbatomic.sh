#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in base noatomic base noatomic; do
  if [ $v == noatomic ]; then export BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_noatomic.so; else unset BEER_HIP_LIB; fi
  timeout 300 python bench.py --config5-only --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', [round(x*1e3,2) for x in d['epoch_s']], {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
done
