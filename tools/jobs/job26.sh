#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j26; mkdir -p $O
for gb in 6 24; do for mf in 2097152 4194304; do
BEER_SCRATCH_GB=$gb python bench.py --config 3 --cov full --frames 4000000 --max-frames $mf --no-cpu-baseline --steps 3 --warmup 1 > $O/full_${gb}_$mf.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/full_${gb}_$mf.json').read().strip().splitlines()[-1]); print('full', $gb, $mf, round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
BEER_SCRATCH_GB=$gb python bench.py --config 3 --max-frames $mf --no-cpu-baseline --steps 3 --warmup 1 > $O/diag_${gb}_$mf.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/diag_${gb}_$mf.json').read().strip().splitlines()[-1]); print('diag', $gb, $mf, round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
done; done
