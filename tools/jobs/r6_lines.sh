#!/bin/bash
# the bench lines alone (no profiler passes): refreshes gpurun_out/r6prof/bench*.json on the final code
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6prof; mkdir -p $O
line() { local n=$1; shift; timeout 900 "$@" 2> $O/$n.stderr | grep "^{" > $O/$n.json; cp bench_detail.json $O/${n}_detail.json 2>/dev/null; }
line bench python bench.py --steps 20 --warmup 5
line bench_c3 python bench.py --config 3 --steps 6 --warmup 2
line bench_c3full python bench.py --config 3 --cov full --frames 2000000 --no-cpu-baseline --steps 6 --warmup 2
timeout 420 python bench.py --config4-only > $O/bench_c4.json 2>>$O/err.log
timeout 420 python bench.py --config5-only > $O/bench_c5.json 2>>$O/err.log
tail -c 1500 $O/bench.json
