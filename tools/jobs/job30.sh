#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "more_than_256 or matrix_core_path" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
