#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j32; mkdir -p $O
rocprofv3 --kernel-trace --stats -f csv -d $O/st -- python bench.py --no-cpu-baseline --no-exact > $O/c2.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, d.get('elbo_rel_err_vs_cpu_fp64'), d.get('stats_rel_err_vs_cpu_fp64'))"
grep -E "pack16|absmax|scale_kernel|llh16|acc16d|unpack|nw_|kl_|Memset|fill" $O/st/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
find $O/st -name '*kernel_trace.csv' -delete
