#!/bin/bash
# fused-accumulation ablations (tools/ab_build.sh af<N> estep_bf16 -DBEER_AF_ABL=<N>)
cd "$GRAFT_REPO_ROOT"
for v in base "$@"; do
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=build_ab/libbeer_hip_$v.so; fi
  python bench.py --config 3 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), {k:round(v['ms'],2) for k,v in d['kernels'].items()})"
done
