#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; rm -rf $O; mkdir -p $O
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-exact --no-check"
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/tr -- $C2 > $O/tr.log 2>&1
python tools/trace_gaps.py $O/tr llhx_kernel 8 > $O/gaps.txt 2>&1
cat $O/gaps.txt | head -60
find $O/tr -name '*kernel_trace.csv' -delete
