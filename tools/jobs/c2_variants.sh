#!/bin/bash
# config-2 step and kernel times with A/B libraries (build_ab/libbeer_hip_<name>.so) and env settings: name[:ENV=VAL,..]
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [[ $spec == *:* ]] && envs=${spec#*:}
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=build_ab/libbeer_hip_$v.so; fi
  env ${envs//,/ } timeout 150 python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-exact --no-check --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$spec', round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
done
