cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "captured" 2>&1 | tail -30 > gpurun_out/t1.log
python - > gpurun_out/c1.log 2>&1 <<'PY'
import json, sys, argparse
sys.argv=['bench.py']
import bench, torch
args=argparse.Namespace(no_cpu_baseline=False)
print(json.dumps(bench.run_config1(args, torch.device('cuda:0')), indent=1))
PY
