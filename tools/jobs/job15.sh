#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j15; mkdir -p $O
cp build_ab/bench_r1.py bench_r1.py
for rep in 1 2; do
python bench_r1.py --steps 20 --warmup 5 --no-cpu-baseline > $O/old_$rep.json 2>$O/old.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact --no-check > $O/new_$rep.json 2>$O/new.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/newfull_$rep.json 2>$O/newfull.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j15/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $O/old.err
