#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c5g; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/tr -- python bench.py --config5-only --no-cpu-baseline > $O/tr.log 2>&1
python tools/trace_gaps.py $O/tr lnfi_kernel 5 > $O/gaps.txt 2>&1
head -50 $O/gaps.txt
find $O/tr -name '*kernel_trace.csv' -delete
