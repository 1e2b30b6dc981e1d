#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j19; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --config 3 --no-cpu-baseline --steps 4 --warmup 2 > $O/c3.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
python bench.py --no-cpu-baseline > $O/c2.json 2>>$O/err.log
python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],3), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
