#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or c3_real or pack_resps or fused or mixtureset or hmm" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 3 --warmup 1 --no-cpu-baseline"
$C3F > $O/c3full.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3full.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/c3.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],2), {k:(round(v['ms'],3),v['launches']) for k,v in d['kernels'].items()})"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc2 -- $C3F > $O/pmc2.log 2>&1
for f in $(find $O/pmc2 -name '*counter_collection.csv'); do
  grep -E "Kernel_Name|llh16_kernel|acc16d|fb_wave_kernel" $f > $f.tmp; mv $f.tmp $f
done
find $O/pmc2 -name '*kernel_trace.csv' -delete
