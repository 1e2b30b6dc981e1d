#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j13; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^E  |AssertionError|passed|failed|Error|^tests.*(FAILED|ERROR)|^FAILED|^ERROR" $O/pytest.log | head -60
