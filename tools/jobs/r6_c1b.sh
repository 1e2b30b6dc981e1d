#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash tools/jobs/r6_tests.sh
python - <<'PY'
import json, sys, argparse
sys.argv=['bench.py']
import bench, torch
args=argparse.Namespace(no_cpu_baseline=True, no_extras=False)
d=bench.run_config1(args, torch.device('cuda:0'))
print({k: round(d[k]['us_per_iteration'],1) for k in ('eager','default','captured')}, d['captured']['mode'], d['captured']['last_elbo'], d['eager']['last_elbo'])
PY
