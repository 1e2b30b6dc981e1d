#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j17; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $O/pmc1 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-check > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -f csv -d $O/pmc2 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-check > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc3 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-check > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $O/pmc4 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-check > $O/pmc4.log 2>&1
python - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/j17/pmc*/pmc_counter_collection.csv')):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        k='acc16d' if 'acc16d' in n else ('llh16' if 'llh16' in n else None)
        if k: per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in per.items():
        out={}
        for n,v in c.items():
            big=max(v); full=[x for x in v if x>.5*big]
            out[n]=round(sum(full)/len(full)/1e6,2)
        print(k,out)
PY
