#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j5; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_accumulation" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for ntc in 4 8; do
BEER_ACCF_NTC=$ntc rocprofv3 --kernel-trace --stats -f csv -d $O/prof_hmm$ntc -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 5 > $O/hmm$ntc.json 2>$O/hmm$ntc.err
cat $O/hmm$ntc.json; head -6 $O/prof_hmm$ntc/hmm_kernel_stats.csv | cut -c1-150
done
