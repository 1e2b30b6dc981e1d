#!/bin/bash
# Copy what tools/jobs/job_r<N>_profiles.sh left under gpurun_out/<dir> into profiles/<round>_*
# (kernel-stats CSVs, bench lines, PMC passes summarised by tools/pmc_summary.py, probe outputs).
#   bash tools/collect_profiles.sh r04 gpurun_out/r4prof
set -e
cd "$(dirname "$0")/.."
R=${1:-r04}; O=${2:-gpurun_out/r4prof}
rm -f profiles/${R}_pmc.json
python tools/pmc_summary.py $R "--command=python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config3 --no-config4" $O/c2_pmc1 $O/c2_pmc2 $O/c2_pmc3
python tools/pmc_summary.py $R "--command=python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline" --tag=c3 $O/c3_pmc1 $O/c3_pmc2 $O/c3_pmc3
python tools/pmc_summary.py $R "--command=python bench.py --config 3 --cov full --frames 2000000 --steps 3 --warmup 2 --no-cpu-baseline" --tag=c3full $O/c3full_pmc1 $O/c3full_pmc2 $O/c3full_pmc3
python tools/pmc_summary.py $R "--command=python tools/probes/c4_prior_path.py full 4" --tag=c4 $O/c4_pmc1 $O/c4_pmc2 $O/c4_pmc3
cp $(ls -t $O/c4_stats/*/*kernel_stats.csv | head -1) profiles/${R}_config4_prior_path_kernel_stats.csv
cp $(ls -t $O/c4bench_stats/*/*kernel_stats.csv | head -1) profiles/${R}_bench_config4_kernel_stats.csv
cp $(ls -t $O/c2_stats/*/*kernel_stats.csv | head -1) profiles/${R}_bench_kernel_stats.csv
cp $(ls -t $O/c3_stats/*/*kernel_stats.csv | head -1) profiles/${R}_bench_config3_kernel_stats.csv
cp $(ls -t $O/c3full_stats/*/*kernel_stats.csv | head -1) profiles/${R}_bench_config3_full_kernel_stats.csv
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_c3.json profiles/${R}_bench_config3.json
cp $O/bench_c3full.json profiles/${R}_bench_config3_full.json
for f in bench_c4.json bench_g2_gloo.json bench_g2_gloo_config3.json chain_len.json err_diag.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/${R}_$f
done
ls -la profiles/${R}_*
