#!/bin/bash
# Round 6, first GPU call: the parity suite with the LDS-staged K1 as the real default, the
# driver's own bench command, what the box offers for reading clocks.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.stdout 2> $O/bench.stderr; echo "bench rc $?"
cp bench_detail.json $O/ 2>/dev/null
wc -c $O/bench.stdout; cat $O/bench.stdout
(ls /sys/class/drm/ ; cat /sys/class/drm/card*/device/pp_dpm_sclk; rocm-smi --showclocks; python -c "import amdsmi; print('amdsmi ok')") > $O/clocks.txt 2>&1
tail -20 $O/clocks.txt
