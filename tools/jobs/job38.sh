#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j38; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
