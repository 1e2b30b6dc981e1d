#!/bin/bash
# config 4 alone: the bench line + a kernel trace of the same command
mkdir -p gpurun_out/c4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python bench.py --config4-only > gpurun_out/c4/bench_c4.json 2> gpurun_out/c4/bench_c4.err
tail -c 600 gpurun_out/c4/bench_c4.err
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/c4/prof -- python bench.py --config4-only > gpurun_out/c4/prof.log 2>&1
f=$(find gpurun_out/c4/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/kstats_head.py "$f" 45 > gpurun_out/c4/kernel_stats_head.txt
find gpurun_out/c4/prof -name "*kernel_trace.csv" -delete
find gpurun_out/c4/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete
find gpurun_out/c4/prof -name "*.db" -delete
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c4/bench_c4.json').read().strip().splitlines()[-1])
c=d.get('config4', d)
for cov in ('diagonal','full'):
    for key in ('vae_step','prior_hot_path','dense_route'):
        s=c[cov][key]
        print(cov, key, '%.1f M frames/s %.1f ms'%(s['value']/1e6, s['ms_per_step']))
        for k,v in s['kernels'].items():
            print('    %-34s %.2f ms x %.1f %s'%(k, v['ms'], v['launches_per_step'], ('%.0f TF'%v['tflops']) if 'tflops' in v else ''))
    print(cov, 'roofline', c[cov]['roofline']['kernel'], c[cov]['roofline']['frac'])
PY
