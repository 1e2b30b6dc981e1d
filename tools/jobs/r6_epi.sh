#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python tools/probes/k1_ab.py 2>/dev/null | grep "k1_lds=1\|bit-identical"
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras --no-exact"
for v in new new new; do
  timeout 300 $C2 > /dev/null 2>&1
  python - <<P
import json
d=json.load(open('bench_detail.json'))
print('$v', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()}, d['parity_vs_cpu_fp64']['bf16x3']['stats_rel_err'], d['elbo_rel_err_vs_cpu_fp64'])
P
done
echo "packed or c2_shape or more_than_256 or beyond_64 or fast_path or c3_real" > tools/jobs/k.txt
bash tools/jobs/r6_tests.sh
