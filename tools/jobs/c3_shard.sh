#!/bin/bash
# One 8-way shard of config 3 (1.25 M frames) on one GPU: ms/step, kernel trace -> busy time and idle gaps.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/${1:-c3shard}; rm -rf $O; mkdir -p $O
F=${2:-1250000}
python bench.py --config 3 --frames $F --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'], 'allreduce', d['all_reduce_ms'], 'mstep', d['m_step_ms'], d['m_step'])
tot=0
for k,v in d['kernels'].items():
    per=v['ms']*v['launches']/d['steps']; tot+=per
    print('  %-40s %.3f ms x %d = %.3f ms/step'%(k, v['ms'], v['launches']/d['steps'], per))
print('sum of timed calls per step', tot)
PY
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/trace -- python bench.py --config 3 --frames $F --no-cpu-baseline --steps 6 --warmup 3 > $O/bench_prof.json 2>> $O/bench.err
python tools/trace_gaps.py $O/trace ${3:-lnfi_kernel} 5 | tee $O/gaps.txt
find $O/trace -name '*kernel_trace.csv' -delete
timeout 300 python tools/probes/host_ahead.py diagonal $F 2>&1 | tee $O/host_ahead.txt
