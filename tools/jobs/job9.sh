#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j9; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_accumulation or c3_real" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "AssertionError|passed|failed" $O/pytest.log | tail -8
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 5 > $O/hmm.json 2>$O/hmm.err
cat $O/hmm.json; head -7 $O/prof/hmm_kernel_stats.csv | cut -c1-150
