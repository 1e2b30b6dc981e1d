#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j7; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_accumulation" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for dbg in 0 3; do
BEER_ACCF_DBG=$dbg rocprofv3 --kernel-trace --stats -f csv -d $O/prof$dbg -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 3 > $O/hmm$dbg.json 2>$O/hmm$dbg.err
echo "dbg=$dbg $(grep accf_kernel $O/prof$dbg/hmm_kernel_stats.csv | cut -d, -f2-4)"
done
cat $O/hmm0.json
