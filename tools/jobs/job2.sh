#!/bin/bash
# round 2, GPU job 2: one-wave-per-utterance forward-backward
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j2; mkdir -p $O
python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
for cov in diagonal; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof_hmm_$cov -o hmm -- python tools/bench_hmm.py --cov $cov --steps 5 --ali-utts 3000 > $O/hmm_$cov.json 2>$O/hmm_$cov.err
  cat $O/hmm_$cov.json; tail -3 $O/hmm_$cov.err
  head -12 $O/prof_hmm_$cov/hmm_kernel_stats.csv | cut -c1-150
done
python tools/bench_hmm.py --cov diagonal --steps 5 > $O/hmm_diag_noprof.json 2>&1; cat $O/hmm_diag_noprof.json
