#!/bin/bash
# Copy what tools/jobs/job_r<N>_profiles.sh left under gpurun_out/<dir> into profiles/<round>_*
# (kernel-stats CSVs, launch statistics without warm-ups, bench lines, PMC passes summarised by
# tools/pmc_summary.py).
#   bash tools/collect_profiles.sh r05 gpurun_out/r5prof
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}; O=${2:-gpurun_out/r6prof}
rm -f profiles/${R}_pmc.json
C2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-extras"
C3="python bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
C3F="python bench.py --config 3 --cov full --frames 2000000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
python tools/pmc_summary.py $R "--command=$C2" --skip=5/25 $O/c2_pmc1 $O/c2_pmc2 $O/c2_pmc3
python tools/pmc_summary.py $R "--command=$C3" --tag=c3 --skip=2/8 $O/c3_pmc1 $O/c3_pmc2 $O/c3_pmc3
python tools/pmc_summary.py $R "--command=$C3F" --tag=c3full --skip=2/8 $O/c3full_pmc1 $O/c3full_pmc2 $O/c3full_pmc3
python tools/pmc_summary.py $R "--command=python tools/probes/c4_prior_path.py full 5" --tag=c4 --skip=2/7 $O/c4_pmc1 $O/c4_pmc2 $O/c4_pmc3
python tools/pmc_summary.py $R "--command=python tools/probes/c4_prior_path.py diagonal 5" --tag=c4d --skip=2/7 --min-read=accd_kernel:736000000 $O/c4d_pmc1 $O/c4d_pmc2 $O/c4d_pmc3
cp $O/kernel_times.json profiles/${R}_kernel_times.json
for n in c2:bench c3:bench_config3 c3full:bench_config3_full c4:config4_prior_path c4d:config4_prior_path_diagonal c5:bench_config5 c4bench:bench_config4; do
  a=${n%%:*}; b=${n#*:}
  f=$(ls -t $O/${a}_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f profiles/${R}_${b}_kernel_stats.csv
  [ -s $O/${a}_timed_stats.csv ] && cp $O/${a}_timed_stats.csv profiles/${R}_${b}_timed_stats.csv
done
# (--profiles-only: called by the job itself ON THE GPU BOX before it runs the bench lines, so that a
#  line's frac_profiled / traffic are priced on the profile of the very box it runs on)
[ "$3" = "--profiles-only" ] && exit 0
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_c3.json profiles/${R}_bench_config3.json
cp $O/bench_c3full.json profiles/${R}_bench_config3_full.json
for n in bench:bench bench_c3:bench_config3 bench_c3full:bench_config3_full bench_g2_gloo:bench_g2_gloo bench_g8_gloo_config3:bench_g8_gloo_config3; do
  a=${n%%:*}; b=${n#*:}
  [ -s $O/${a}_detail.json ] && cp $O/${a}_detail.json profiles/${R}_${b}_detail.json
done
for f in bench_c4.json bench_c5.json bench_g2_gloo.json bench_g8_gloo_config3.json; do
  [ -s $O/$f ] && cp $O/$f profiles/${R}_$f
done
ls -la profiles/${R}_*
