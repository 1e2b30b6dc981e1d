#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j40; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or c3_real or more_than_256 or phone" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()}, d['count_conservation_rel_err'])"
