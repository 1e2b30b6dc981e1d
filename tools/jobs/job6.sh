#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j6; mkdir -p $O
for dbg in 0 1 2 3 4 7; do
BEER_ACCF_DBG=$dbg rocprofv3 --kernel-trace --stats -f csv -d $O/prof$dbg -o hmm -- python tools/bench_hmm.py --cov diagonal --steps 3 > $O/hmm$dbg.json 2>$O/hmm$dbg.err
echo "dbg=$dbg $(grep accf_kernel $O/prof$dbg/hmm_kernel_stats.csv | cut -d, -f2-4)"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $O/pmc1 -o pmc -- python tools/bench_hmm.py --cov diagonal --steps 2 > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -f csv -d $O/pmc2 -o pmc -- python tools/bench_hmm.py --cov diagonal --steps 2 > $O/pmc2.log 2>&1
python - <<'PY'
import csv,collections
for f in ['gpurun_out/j6/pmc1/pmc_counter_collection.csv','gpurun_out/j6/pmc2/pmc_counter_collection.csv']:
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'accf_kernel' in r['Kernel_Name']: per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in per.items(): print(k, sum(v)/len(v), len(v))
PY
