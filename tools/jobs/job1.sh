#!/bin/bash
# round 2, GPU job 1: new parity tests, K1 stride A/B, C3 kernel breakdown
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j1; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for rep in 1 2; do
BEER_HIP_LIB=$PWD/build_ab/libbeer_hip_r1.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r1_$rep.json 2>$O/bench_r1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_new_$rep.json 2>$O/bench_new.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})
    except Exception as e: print(f,'ERR',e)
PY
# C3 breakdown (kernel trace)
for cov in diagonal full; do
  rocprofv3 --kernel-trace --stats -f csv -d $O/prof_hmm_$cov -o hmm -- python tools/bench_hmm.py --cov $cov --steps 3 > $O/hmm_$cov.json 2>$O/hmm_$cov.err
  cat $O/hmm_$cov.json
  f=$(ls $O/prof_hmm_$cov/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $O/prof_hmm_$cov/*kernel_stats.csv | head -1)
  head -14 $f | cut -c1-160
done
# PMC: LDS conflicts of K1 (new lib)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $O/pmc_new -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_new.log 2>&1
ls -R $O/pmc_new | head
