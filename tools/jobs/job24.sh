#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j24; mkdir -p $O
for v in base nofold nostore; do
  if [ $v == base ]; then unset BEER_HIP_LIB; else export BEER_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libbeer_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats -f csv -d $O/$v -- python bench.py --config 3 --cov full --frames 1000000 --no-cpu-baseline --steps 3 --warmup 1 > $O/$v.log 2>&1
  find $O/$v -name '*kernel_trace.csv' -delete
  echo $v; grep -E "acc16d|llh16|gt_image|fb_wave" $O/$v/*/*kernel_stats.csv | cut -d, -f2-4 | head -5
done
