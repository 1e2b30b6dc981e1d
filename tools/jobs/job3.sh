#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j3; mkdir -p $O
python -m pytest tests -m gpu -q --durations=5 --deselect "tests/test_gpu_parity.py::test_bench_kernel_variant_vs_oracle[diagonal-256]" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
rocprofv3 --kernel-trace --stats -f csv -d $O/prof_hmm -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 5 --ali-utts 3000 > $O/hmm.json 2>$O/hmm.err
cat $O/hmm.json; head -8 $O/prof_hmm/hmm_kernel_stats.csv | cut -c1-140
