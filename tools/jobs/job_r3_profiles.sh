#!/bin/bash
# Round-3 profiles: kernel-trace stats + three PMC passes of the bench commands
# (config 2 alone, config 3 diagonal, config 3 full), then the driver's line itself.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3prof; rm -rf $O; mkdir -p $O
T="timeout 600"
C2="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config3"
C3="python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline"
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 3 --warmup 1 --no-cpu-baseline"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KEEP="Kernel_Name|llhx_kernel|accx_kernel|accf_kernel|accfi_kernel|frame_image_kernel|fb_wave_kernel|llh_kernel|acc_kernel|gt_image|xt_image"
run() {  # name, command
  $T rocprofv3 --kernel-trace --stats -f csv -d $O/$1_stats -- $2 > $O/$1_stats.log 2>&1
  $T rocprofv3 --kernel-trace --pmc $P1 -f csv -d $O/$1_pmc1 -- $2 > $O/$1_pmc1.log 2>&1
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/$1_pmc2 -- $2 > $O/$1_pmc2.log 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/$1_pmc3 -- $2 > $O/$1_pmc3.log 2>&1
  # keep what is judged small: drop the raw kernel traces, keep stats + counters
  find $O/$1_stats -name '*kernel_trace.csv' -delete
  for i in 1 2 3; do find $O/$1_pmc$i -name '*kernel_trace.csv' -delete
    for f in $(find $O/$1_pmc$i -name '*counter_collection.csv'); do
      grep -E "$KEEP" $f > $f.tmp; mv $f.tmp $f
    done
  done
}
run c2 "$C2"
run c3 "$C3"
run c3full "$C3F"
du -sh $O; find $O -name '*.csv' | head -40
# the driver's line (with cpu baselines and the config-3 sub-objects), and the config-3 lines alone
$T python bench.py --steps 20 --warmup 5 > $O/bench.json 2>>$O/err.log
$T python bench.py --config 3 > $O/bench_c3.json 2>>$O/err.log
$T python bench.py --config 3 --cov full --frames 2000000 --no-cpu-baseline > $O/bench_c3full.json 2>>$O/err.log
tail -c 1500 $O/bench.json
