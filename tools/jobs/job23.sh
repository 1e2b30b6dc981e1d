#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or c3_real or pack_resps" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
python bench.py --config 3 --cov full --frames 1000000 --no-cpu-baseline --steps 3 --warmup 1 > $O/c3full.json 2>$O/err.log
tail -3 $O/err.log
python -c "
import json; d=json.loads(open('$O/c3full.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
