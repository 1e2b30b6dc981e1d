#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j8; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_accumulation or c3_real" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "AssertionError|passed|failed" $O/pytest.log | tail -8
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $O/pmc1 -o pmc -- python tools/bench_hmm.py --cov diagonal --steps 2 > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -f csv -d $O/pmc2 -o pmc -- python tools/bench_hmm.py --cov diagonal --steps 2 > $O/pmc2.log 2>&1
python - <<'PY'
import csv,collections
for f in ['gpurun_out/j8/pmc1/pmc_counter_collection.csv','gpurun_out/j8/pmc2/pmc_counter_collection.csv']:
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        k='accf' if 'accf_kernel' in n else ('llh16' if 'llh16' in n else ('fbwave' if 'fb_wave' in n else None))
        if k: per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in per.items():
        print(k, {n: round(sum(v)/len(v)/1e6,1) for n,v in c.items()})
PY
