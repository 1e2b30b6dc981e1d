#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j12; mkdir -p $O
( time python bench.py ) > $O/bench_c2.json 2>$O/bench_c2.err; tail -3 $O/bench_c2.err
( time python bench.py --config 3 ) > $O/bench_c3.json 2>$O/bench_c3.err; tail -3 $O/bench_c3.err
BEER_BENCH_BACKEND=gloo python bench.py --gpus 2 --config 3 --frames 2000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c3_gloo2.json 2>$O/bench_c3_gloo2.err; tail -2 $O/bench_c3_gloo2.err
BEER_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --frames 262144 --no-cpu-baseline --no-check > $O/bench_c2_gloo2.json 2>$O/bench_c2_gloo2.err; tail -2 $O/bench_c2_gloo2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/j12/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'value', round(d['value']/1e6,1),'M f/s', 'ms', round(d['ms_per_step'],3), 'n', d['n_gpus'], d['scaling'], 'ar', round(d['all_reduce_ms'],3), 'm', round(d['m_step_ms'],3))
        print('   roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k!='note'})
        print('   kernels', {k:round(v['ms'],3) for k,v in d['kernels'].items()})
        for k in ('f32_exact','cpu_baseline','elbo_rel_err_vs_cpu_fp64','stats_rel_err_vs_cpu_fp64'):
            if k in d: print('   ',k, d[k] if not isinstance(d[k],dict) else {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items() if a not in('note',)})
    except Exception as e: print(f,'ERR',e, open(f).read()[-300:])
PY
