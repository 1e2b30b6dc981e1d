#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hmm.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 3 6 12; do
BEER_ACCF_ROUNDS=$r python bench.py --config 3 --frames 4000000 --no-cpu-baseline --steps 4 --warmup 2 > $O/c3_$r.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3_$r.json').read().strip().splitlines()[-1]); print($r, round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
done
C3="python bench.py --config 3 --frames 4000000 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc2 -- $C3 > $O/pmc2.log 2>&1
for f in $(find $O/pmc2 -name '*counter_collection.csv'); do
  grep -E "Kernel_Name|llh16_kernel|accf_kernel|fb_wave_kernel" $f > $f.tmp; mv $f.tmp $f
done
find $O/pmc2 -name '*kernel_trace.csv' -delete
