#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c5; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --config5-only > $O/c5.json 2> $O/c5.err; echo rc $?
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -- python bench.py --config5-only --no-cpu-baseline > $O/stats.log 2>&1
find $O/stats -name '*kernel_trace.csv' -delete
python - <<'P'
import json,glob,csv
d=json.load(open('gpurun_out/r6c5/c5.json'))
for k in ('value','wall_s','stages_s','epoch_s','training_frames_per_s','gaussians','kernels','frames'):
    print(k, d.get(k))
print(d.get('cpu_baseline',{}).get('value'))
f=glob.glob('gpurun_out/r6c5/stats/*/*kernel_stats.csv')[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<14: print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
P
