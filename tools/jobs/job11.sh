#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j11; mkdir -p $O
python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_parity.py::test_bench_kernel_variant_vs_oracle" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "AssertionError|passed|failed|Error" $O/pytest.log | tail -8
python tools/bench_hmm.py --cov diagonal --steps 10 > $O/hmm_noprof.json 2>&1; cat $O/hmm_noprof.json
rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 5 > $O/hmm.json 2>$O/hmm.err
head -7 $O/prof/hmm_kernel_stats.csv | cut -c1-150
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
