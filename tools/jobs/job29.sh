#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense_graphs or long_chains or hmm or viterbi or g04 or g07 or phone" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
