#!/bin/bash
# Where a config-3 iteration's wall time goes: kernel trace -> busy time per kernel and the idle gaps
# (tools/trace_gaps.py), and whether the host keeps ahead of the device (tools/probes/host_ahead.py).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/${1:-c3gaps}; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/trace -- python bench.py --config 3 --no-cpu-baseline --steps 3 --warmup 2 > $O/bench.json 2> $O/bench.err
python tools/trace_gaps.py $O/trace ${2:-lnfi_kernel} 7 | tee $O/gaps.txt
find $O/trace -name '*kernel_trace.csv' -delete
timeout 300 python tools/probes/host_ahead.py diagonal 10000000 2>&1 | tee $O/host_ahead.txt
