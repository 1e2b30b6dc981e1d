#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j16; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or bench_kernel or full_size or c2_shape" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^E  |passed|failed|^FAILED|^ERROR" $O/pytest.log | head -20
for rep in 1 2; do
for form in 1 2; do
BEER_ACC16P=$form python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact --no-check > $O/form${form}_$rep.json 2>$O/err.log
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j16/form*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})
    except Exception as e: print(f,'ERR',e)
PY
