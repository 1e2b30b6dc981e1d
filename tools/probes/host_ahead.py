"""Does the host keep ahead of the device in a config-3 iteration?  Per step: the host time
to issue it (no synchronisation) against the device time it takes (events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
from beer_amd.distributed import all_reduce_elbo

dev = torch.device('cuda:0')
cov = sys.argv[1] if len(sys.argv) > 1 else 'diagonal'
lengths = bench.hmm_corpus(int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000)
X = torch.randn(sum(lengths), bench.D, device=dev)
ploop = bench.make_phone_loop(cov, dev)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1., graph=len(sys.argv) > 3)
marks = []
def step():
    t = [time.perf_counter()]
    optim.init_step()
    t.append(time.perf_counter())
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    t.append(time.perf_counter())
    elbo, _ = all_reduce_elbo(elbo, ploop, len(lengths))
    t.append(time.perf_counter())
    elbo.backward()
    t.append(time.perf_counter())
    optim.step()
    t.append(time.perf_counter())
    marks.append(t)
    return elbo
for _ in range(2):
    step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
marks.clear()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
t0 = time.perf_counter()
for i in range(5):
    ev[i].record()
    step()
ev[5].record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'5 steps: host issue {t_issue * 1e3:.1f} ms, until device done {t_all * 1e3:.1f} ms; device per step',
      [round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(5)])
names = ['init_step', 'accumulate_elbo', 'all_reduce', 'backward', 'optim.step']
for k, n in enumerate(names):
    print(f'   host {n:16s}', [round((m[k + 1] - m[k]) * 1e3, 2) for m in marks])
