#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j14; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^E  |passed|failed|^FAILED|^ERROR" $O/pytest.log | head -40
python tools/bench_hmm.py --cov diagonal --steps 5 --ncomp 10 > $O/hmm_g10.json 2>&1; tail -1 $O/hmm_g10.json
