#!/bin/bash
# PMC passes over the fused forward-backward launch alone (tools/probes/fb_time.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/fbpmc; rm -rf $O; mkdir -p $O
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for i in 1 2; do
  eval P=\$P$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $O/p$i -- python tools/probes/fb_time.py 3333334 3 > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for i in (1, 2):
    f = glob.glob(f'gpurun_out/fbpmc/p{i}/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv', i); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'fb_wave_kernel' in r['Kernel_Name'] and 'log' not in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        print(f'{k:24s} {sum(v) / len(v):16.0f}  per frame {sum(v) / len(v) / 3333586:10.2f}')
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete
