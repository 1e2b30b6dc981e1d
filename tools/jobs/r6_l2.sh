#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6l2; rm -rf $O; mkdir -p $O
C3="env BEER_ACCFI_PERSIST=1 python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -f csv -d $O/p1 -- $C3 > $O/p1.log 2>&1
python - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/r6l2/p1/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    for k in ('accfi_kernel','lnfi_kernel','fb_wave_kernel'):
        if k in n: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c:(len(x), max(x)) for c,x in v.items()})
P
find $O -name '*.csv' -size +2M -delete
tail -3 $O/p1.log
