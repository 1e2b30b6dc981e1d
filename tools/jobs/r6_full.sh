#!/bin/bash
# the whole GPU suite, then the driver's bench command
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; rm -rf $O; mkdir -p $O
bash tools/jobs/r6_tests.sh
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.stdout 2> $O/bench.stderr; echo "bench rc $?"
cp bench_detail.json $O/ 2>/dev/null
wc -c $O/bench.stdout; cat $O/bench.stdout; tail -3 $O/bench.stderr | cut -c1-300
