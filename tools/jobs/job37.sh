#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j37; mkdir -p $O
for v in 1 0 1 0; do
BEER_KL_SIDE_STREAM=$v python bench.py --no-cpu-baseline --no-exact --steps 30 --warmup 5 > $O/c2_$v.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c2_$v.json').read().strip().splitlines()[-1]); print($v, round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, d.get('elbo_rel_err_vs_cpu_fp64'))"
done
