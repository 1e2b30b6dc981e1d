#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6accd; rm -rf $O; mkdir -p $O
timeout 300 python tools/probes/c4_acc_ab.py > $O/ab.txt 2>&1; head -12 $O/ab.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $O/pmc -- python tools/probes/c4_prior_path.py diagonal 3 > $O/pmc.log 2>&1
python - <<'P'
import csv,glob,collections
f=glob.glob('gpurun_out/r6accd/pmc/*/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'accd_kernel' in r['Kernel_Name']:
        agg['accd'][r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
for k,v in agg['accd'].items(): print(k, v/cnt[k])
P
find $O/pmc -name '*.csv' -size +1M -delete
bash tools/jobs/r6_tests.sh
