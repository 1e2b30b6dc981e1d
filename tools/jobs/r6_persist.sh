#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6p; rm -rf $O; mkdir -p $O
bash tools/jobs/r6_tests.sh
for p in 0 1 0 1; do
  BEER_ACCFI_PERSIST=$p timeout 600 python bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/c3_$p.json 2> $O/c3_$p.err
  python - <<P
import json
d=json.load(open('gpurun_out/r6p/c3_$p.json')); dd=json.load(open('bench_detail.json'))
print('persist=$p', round(d['ms_per_step'],2), 'ms/step', {k:round(v['ms'],2) for k,v in dd['kernels'].items()}, dd.get('count_conservation_rel_err'))
P
done
