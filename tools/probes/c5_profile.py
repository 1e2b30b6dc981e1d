"""Host profile of one config-5 training epoch (alignment graphs)."""
import cProfile, pstats, sys, os, argparse, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('BEER_BENCH_C5_HOURS', '1.0')
sys.argv = ['bench.py']
import torch, bench
args = argparse.Namespace(no_cpu_baseline=True)
pr = cProfile.Profile()
buf = io.StringIO()
pr.enable()
out = bench.run_config5(args, torch.device('cuda:0'))
pr.disable()
print(json.dumps({k: out[k] for k in ('stages_s', 'epoch_s', 'training_frames_per_s', 'value')}))
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
