#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j22; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --config 3 --cov full --frames 1000000 --no-cpu-baseline --steps 3 --warmup 1 > $O/c3full.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3full.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
rocprofv3 --kernel-trace --stats -f csv -d $O/full_stats -- python bench.py --config 3 --cov full --frames 1000000 --no-cpu-baseline --steps 3 --warmup 1 > $O/full_stats.log 2>&1
find $O/full_stats -name '*kernel_trace.csv' -delete
head -12 $O/full_stats/*/*kernel_stats.csv | cut -c1-180
