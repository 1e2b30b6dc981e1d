#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/c4gaps; rm -rf $O; mkdir -p $O
timeout 300 python tools/probes/c4_prior_path.py full 6
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/full -- python tools/probes/c4_prior_path.py full 6 > $O/full.log 2>&1
python tools/trace_gaps.py $O/full sgrad_kernel 5
head -30 $(find $O/full -name '*kernel_stats.csv' | head -1) | cut -c1-150
find $O -name '*kernel_trace.csv' -delete
