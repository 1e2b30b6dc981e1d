#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/tr -- python tools/probes/c1_trace.py > $O/tr.log 2>&1
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r6c1/tr/*/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    c=int(r['Calls'])
    if c>=50:
        print(r['Name'][:95], c, round(float(r['AverageNs'])/1e3,2)); tot+=float(r['TotalDurationNs'])/c if c in (58,59,60,116,118,120,174,177,180) else 0
print('kernel us per iteration (approx)', tot/1e3)
P
find $O/tr -name '*kernel_trace.csv' -delete
