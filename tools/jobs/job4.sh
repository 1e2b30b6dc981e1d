#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j4; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_accumulation or c3_ or phoneloop or g9 or packed_path" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_hmm -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 5 > $O/hmm.json 2>$O/hmm.err
cat $O/hmm.json; tail -3 $O/hmm.err; head -9 $O/prof_hmm/hmm_kernel_stats.csv | cut -c1-150
