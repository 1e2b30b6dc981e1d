import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from beer_amd import hmm_kernels as hk, kernels, graph
from beer_amd.inference import batch as B
from beer_amd.models import parameters, basemodel, sequence
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
lengths = []
while sum(lengths) < 1000000: lengths.append(int(rng.randint(200, 401)))
X = torch.randn(sum(lengths), 40, device=dev)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
log = []
def wrap(obj, name, tag=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((tag or name, t0, time.perf_counter())); return r
    setattr(obj, name, w)
wrap(hk, 'HmmBatch'); wrap(B, '_hmm_batch'); wrap(sequence.PhoneLoop, 'phone_counts'); wrap(basemodel.Model, 'kl_div_posterior_prior', 'KL')
wrap(parameters.ConjugateBayesianParameter, 'natural_grad_update', 'ngu'); wrap(sequence.PhoneLoop, '_on_weights_update', 'wupd')
wrap(graph.DeviceGraph, '__init__', 'DevGraph'); wrap(B, '_emission_estep'); wrap(hk, 'forward_backward', 'fb'); wrap(kernels, 'normal_accumulate', 'acc')
wrap(B, 'pack_utterances', 'pack'); wrap(B, '_sub_batches'); wrap(B, '_groups'); wrap(B, '_finish')
def run():
    ta = time.perf_counter(); optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    tb = time.perf_counter(); elbo.backward(); optim.step(); log.append(('ITER', ta, time.perf_counter())); log.append(('tail', tb, time.perf_counter()))
for _ in range(3): run()
torch.cuda.synchronize(); log.clear()
T0 = time.perf_counter()
for i in range(6): run()
torch.cuda.synchronize()
print('total per iter', (time.perf_counter() - T0) / 6 * 1e3)
for n, a, b in sorted(log, key=lambda x: x[1]):
    if b - a > 0.0008 or n in ('ITER',):
        print('%8.1f %-12s %7.1f ms' % ((a - T0) * 1e3, n, (b - a) * 1e3))
