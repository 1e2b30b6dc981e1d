#!/bin/bash
# Round-2 profiles: kernel-trace stats + three PMC passes of the two bench commands.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j27; rm -rf $O; mkdir -p $O
C2="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
C3="python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {  # name, command
  rocprofv3 --kernel-trace --stats -f csv -d $O/$1_stats -- $2 > $O/$1_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc $P1 -f csv -d $O/$1_pmc1 -- $2 > $O/$1_pmc1.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/$1_pmc2 -- $2 > $O/$1_pmc2.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/$1_pmc3 -- $2 > $O/$1_pmc3.log 2>&1
  # keep what is judged small: drop the raw kernel traces, keep stats + counters
  find $O/$1_stats -name '*kernel_trace.csv' -delete
  for i in 1 2 3; do find $O/$1_pmc$i -name '*kernel_trace.csv' -delete
    for f in $(find $O/$1_pmc$i -name '*counter_collection.csv'); do
      grep -E "Kernel_Name|llh16_kernel|acc16|accf_kernel|fb_wave_kernel|llh_kernel|acc_kernel|gt_image" $f > $f.tmp; mv $f.tmp $f
    done
  done
}
run c2 "$C2"
run c3 "$C3"
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 3 --warmup 1 --no-cpu-baseline"
run c3full "$C3F"
du -sh $O; find $O -name '*.csv' | head -30
tail -1 $O/c2_stats.log; tail -1 $O/c3_stats.log; tail -1 $O/c3full_stats.log
# the bench lines themselves (with cpu baselines)
python bench.py > $O/bench_c2.json 2>>$O/err.log; python bench.py --config 3 > $O/bench_c3.json 2>>$O/err.log
python bench.py --config 3 --cov full --frames 2000000 --no-cpu-baseline > $O/bench_c3full.json 2>>$O/err.log
tail -c 600 $O/bench_c2.json; tail -c 400 $O/bench_c3.json
