#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6rot; rm -rf $O; mkdir -p $O
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras --no-exact"
for v in norot rot norot rot norot rot; do
  if [ $v == norot ]; then export BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_norot.so; else unset BEER_HIP_LIB; fi
  timeout 300 $C2 > $O/$v.json 2>/dev/null
  python - <<P
import json
d=json.load(open('bench_detail.json'))
print('$v', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()}, d['parity_vs_cpu_fp64']['bf16x3']['stats_rel_err'])
P
done
unset BEER_HIP_LIB
echo "packed or c2_shape or chain_length" > tools/jobs/k.txt
bash tools/jobs/r6_tests.sh
