#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j18; mkdir -p $O
for mf in 1048576 2097152 4194304 16777216; do
python bench.py --config 3 --max-frames $mf --no-cpu-baseline --steps 4 --warmup 2 > $O/c3_$mf.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3_$mf.json').read().strip().splitlines()[-1]); print($mf, round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, 'm_step', round(d['m_step_ms'],3))"
done
