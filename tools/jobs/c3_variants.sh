#!/bin/bash
# config-3 kernel times with A/B libraries (build_ab/libbeer_hip_<name>.so) and env settings: name[:ENV=VAL]
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [[ $spec == *:* ]] && envs=${spec#*:}
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=build_ab/libbeer_hip_$v.so; fi
  env ${envs//,/ } timeout 150 python bench.py --config 3 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$spec', round(d['value']/1e6,1), {k:round(v['ms'],2) for k,v in d['kernels'].items()}, 'elbo/frame', d.get('elbo_per_frame'))"
done
