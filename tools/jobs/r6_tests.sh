#!/bin/bash
# pytest -m gpu on the box; an optional -k expression in gpurun_in/k.txt (one line)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; rm -rf $O; mkdir -p $O
if [ -s tools/jobs/k.txt ]; then
  timeout 2400 python -m pytest tests -m gpu -q -k "$(cat tools/jobs/k.txt)" > $O/pytest.log 2>&1
else
  timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
fi
echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc " $O/pytest.log | tail -40
