#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j31; mkdir -p $O
python tools/debug/wide_iso.py 2>&1 | tail -16
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
python bench.py --no-cpu-baseline --no-exact > $O/c2.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, d.get('elbo_rel_err_vs_cpu_fp64'), d.get('stats_rel_err_vs_cpu_fp64'))"
