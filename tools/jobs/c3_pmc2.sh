#!/bin/bash
# Issue / activity counters of the config-3 kernels (lnfi, fb, accfi) beyond the standard passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/c3pmc2; rm -rf $O; mkdir -p $O
CMD="python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
for i in 1 2 3; do
  eval P=\$P$i
  timeout 400 rocprofv3 --kernel-trace --pmc $P -f csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for i in (1, 2, 3):
    f = glob.glob(f'gpurun_out/c3pmc2/p{i}/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv', i); continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        for k in ('lnfi_kernel', 'accfi_kernel', 'fb_wave_kernel<'):
            if k in r['Kernel_Name']:
                per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in per.items():
        for n, v in c.items():
            big = max(v); v = [x for x in v if x > .5 * big]
            print(f'{k:16s} {n:32s} {sum(v) / len(v):16.0f}  per frame {sum(v) / len(v) / 10000004:10.2f}')
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete
