#!/bin/bash
# Round-6 profiles.  Every bench command is profiled AT ITS OWN STEP COUNTS: rocprofv3 kernel
# trace + stats, launch statistics without the warm-up launches (tools/trace_stats.py ->
# r06_kernel_times.json, what `roofline.frac_profiled` of a bench line is priced on), three PMC
# passes (--kernel-trace --pmc only).  Then the lines themselves.  Every command under a timeout.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6prof; rm -rf $O; mkdir -p $O
T="timeout 420"
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras"
C3="python bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
C4="python tools/probes/c4_prior_path.py full 5"
C4D="python tools/probes/c4_prior_path.py diagonal 5"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
KEEP="Kernel_Name|llhx_kernel|lnfi_kernel|accx_kernel|accf_kernel|accfi_kernel|frame_image_kernel|fb_wave_kernel|llh_kernel|acc_kernel|accd_kernel|gt_image|xt_image|sgrad_kernel"
run() {  # name, steps, warmup, command
  $T rocprofv3 --kernel-trace --stats -f csv -d $O/$1_stats -- $4 > $O/$1_stats.log 2>&1
  python tools/trace_stats.py $O/$1_stats $O/$1_timed_stats.csv $O/kernel_times.json --tag=$5 --steps=$2 --warmup=$3 "--command=$4" > $O/$1_timed.txt 2>&1
  $T rocprofv3 --kernel-trace --pmc $P1 -f csv -d $O/$1_pmc1 -- $4 > $O/$1_pmc1.log 2>&1
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/$1_pmc2 -- $4 > $O/$1_pmc2.log 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/$1_pmc3 -- $4 > $O/$1_pmc3.log 2>&1
  find $O/$1_stats -name '*kernel_trace.csv' -delete
  for i in 1 2 3; do find $O/$1_pmc$i -name '*kernel_trace.csv' -delete
    for f in $(find $O/$1_pmc$i -name '*counter_collection.csv'); do
      grep -E "$KEEP" $f > $f.tmp; mv $f.tmp $f
    done
  done
}
run c2 20 5 "$C2" ""
run c3 6 2 "$C3" c3
run c3full 6 2 "$C3F" c3full
run c4 5 2 "$C4" c4
run c4d 5 2 "$C4D" c4d
$T rocprofv3 --kernel-trace --stats -f csv -d $O/c5_stats -- python bench.py --config5-only --no-cpu-baseline > $O/c5_stats.log 2>&1
find $O/c5_stats -name '*kernel_trace.csv' -delete
$T rocprofv3 --kernel-trace --stats -f csv -d $O/c4bench_stats -- python bench.py --config4-only --no-cpu-baseline > $O/c4bench_stats.log 2>&1
find $O/c4bench_stats -name '*kernel_trace.csv' -delete
du -sh $O
bash tools/collect_profiles.sh r06 gpurun_out/r6prof --profiles-only > $O/collect.log 2>&1
# the lines: the driver's (prices frac_profiled / traffic on profiles/r06_kernel_times.json and
# r06_pmc.json, which the collector call above has just written from THIS box's passes), and the
# workloads alone
# (stdout of a bench run is the compact line; the full objects are bench_detail.json, kept per run)
line() {  # name, command...
  local n=$1; shift
  timeout 900 "$@" 2> $O/$n.stderr | grep "^{" > $O/$n.json    # (gloo prints its connection lines on stdout)
  cp bench_detail.json $O/${n}_detail.json 2>/dev/null
}
line bench python bench.py --steps 20 --warmup 5
line bench_c3 python bench.py --config 3 --steps 6 --warmup 2
line bench_c3full python bench.py --config 3 --cov full --frames 2000000 --no-cpu-baseline --steps 6 --warmup 2
$T python bench.py --config4-only > $O/bench_c4.json 2>>$O/err.log
$T python bench.py --config5-only > $O/bench_c5.json 2>>$O/err.log
BEER_BENCH_BACKEND=gloo line bench_g2_gloo python bench.py --gpus 2 --no-cpu-baseline --steps 5 --warmup 2 --no-config3
BEER_BENCH_BACKEND=gloo line bench_g8_gloo_config3 python bench.py --gpus 8 --config 3 --no-cpu-baseline --steps 3 --warmup 2
tail -c 1200 $O/bench.json; tail -5 $O/err.log
