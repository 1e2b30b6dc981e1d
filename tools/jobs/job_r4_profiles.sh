#!/bin/bash
# Round-4 profiles: kernel-trace stats + three PMC passes of the bench commands (config 2 alone,
# config 3 diagonal, config 3 full), the driver's line itself, the config-3 / config-4 lines alone,
# the 2-rank gloo lines, and the precision probes.  Every command under its own timeout.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4prof; rm -rf $O; mkdir -p $O
T="timeout 300"
C2="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config3 --no-config4"
C3="python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline"
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 3 --warmup 2 --no-cpu-baseline"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C4="python tools/probes/c4_prior_path.py full 4"
KEEP="Kernel_Name|llhx_kernel|lnfi_kernel|accx_kernel|accf_kernel|accfi_kernel|frame_image_kernel|fb_wave_kernel|llh_kernel|acc_kernel|gt_image|xt_image|sgrad_kernel"
run() {  # name, command
  $T rocprofv3 --kernel-trace --stats -f csv -d $O/$1_stats -- $2 > $O/$1_stats.log 2>&1
  $T rocprofv3 --kernel-trace --pmc $P1 -f csv -d $O/$1_pmc1 -- $2 > $O/$1_pmc1.log 2>&1
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/$1_pmc2 -- $2 > $O/$1_pmc2.log 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/$1_pmc3 -- $2 > $O/$1_pmc3.log 2>&1
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
run c4 "$C4"          # the prior hot path of config 4 (full covariance, one-sample route) alone
$T rocprofv3 --kernel-trace --stats -f csv -d $O/c4bench_stats -- python bench.py --config4-only > $O/c4bench_stats.log 2>&1
find $O/c4bench_stats -name '*kernel_trace.csv' -delete
du -sh $O
# the driver's line (cpu baselines, config-3 and config-4 sub-objects), and the lines alone
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>>$O/err.log
$T python bench.py --config 3 --steps 6 --warmup 2 > $O/bench_c3.json 2>>$O/err.log
$T python bench.py --config 3 --cov full --frames 2000000 --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_c3full.json 2>>$O/err.log
$T python bench.py --config4-only > $O/bench_c4.json 2>>$O/err.log
# N = 2 over gloo on the one GPU (the N > 1 code path end to end: spawn, shard, all-reduce, captured M-step)
BEER_BENCH_BACKEND=gloo $T python bench.py --gpus 2 --no-cpu-baseline --steps 5 --warmup 2 --no-config4 > $O/bench_g2_gloo.json 2>>$O/err.log
BEER_BENCH_BACKEND=gloo $T python bench.py --gpus 2 --config 3 --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_g2_gloo_config3.json 2>>$O/err.log
# precision probes
$T python tools/probes/chain_len.py > $O/chain_len.json 2>>$O/err.log
$T python tools/probes/err_diag.py > $O/err_diag.txt 2>>$O/err.log
tail -c 1200 $O/bench.json; tail -5 $O/err.log
