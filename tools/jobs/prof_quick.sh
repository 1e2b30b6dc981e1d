#!/bin/bash
# One or more PMC passes of a bench command, summarised (optimisation loop; not the judged profiles).
#   tools/jobs/prof_quick.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]     (BENCH_CMD overrides)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
CMD=${BENCH_CMD:-"python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exact --no-check"}
i=0
for P in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -f csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1
  find $O/p$i -name '*kernel_trace.csv' -delete
  python tools/pmc_quick.py $O/p$i > $O/p$i.txt 2>&1
  cat $O/p$i.txt
  find $O/p$i -name '*counter_collection.csv' -delete
done
