import sys, time, torch, numpy as np, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from beer_amd import _hip
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
lengths = []
while sum(lengths) < 1000000: lengths.append(int(rng.randint(200, 401)))
X = torch.randn(sum(lengths), 40, device=dev)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
ev = []
orig = _hip.call
def timed(name, *args):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); orig(name, *args); b.record(); ev.append((name, a, b, time.perf_counter() - t0))
_hip.call = timed
for m in (beer.kernels, beer.hmm_kernels): m._hip.call = timed
def run():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    elbo.backward(); optim.step()
    torch.cuda.synchronize()
run(); run(); ev.clear()
t0 = time.perf_counter(); run(); tot = time.perf_counter() - t0
agg = collections.OrderedDict()
for n, a, b, h in ev:
    g = agg.setdefault(n, [0, 0., 0.]); g[0] += 1; g[1] += a.elapsed_time(b); g[2] += h * 1e3
print('iteration ms', tot * 1e3)
for n, g in agg.items(): print(f'{n:32s} n={g[0]:3d} gpu={g[1]:8.2f}ms host={g[2]:8.2f}ms')
_hip.call = orig
for m in (beer.kernels, beer.hmm_kernels): m._hip.call = orig
for i in range(6):
    t0 = time.perf_counter(); run(); print('iter', i, (time.perf_counter() - t0) * 1e3)
def run_nosync():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    elbo.backward(); optim.step()
    return elbo
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5): e = run_nosync()
torch.cuda.synchronize(); print('nosync avg', (time.perf_counter() - t0) / 5 * 1e3)
import cProfile, pstats
for i in range(3):
    pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0; pr.disable()
    print('=== iter', i, dt * 1e3, 'update_count', optim.update_count)
    if dt > 0.035:
        pstats.Stats(pr).sort_stats('tottime').print_stats(12)
