#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j33; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench_kernel_variant or more_than_256 or c3_real or c2_shape" > $O/pytest.log 2>&1; grep -E "rel err|passed|failed|FAILED" $O/pytest.log | head -20
