#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j39; mkdir -p $O
for v in base k1s1 base k1s1; do
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libbeer_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-exact --steps 20 --warmup 5 > $O/c2_$v.json 2>$O/err.log
  python -c "
import json; d=json.loads(open('$O/c2_$v.json').read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, d.get('elbo_rel_err_vs_cpu_fp64'))"
done
